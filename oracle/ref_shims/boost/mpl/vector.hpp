/* ref_shims: the sliver of Boost.MPL that src/libcore/fmtconv.cpp:1171-1208 and src/shapes/ply/ply_parser.hpp:65-175 use (type lists, fold /
 * transform / for_each with placeholder lambda expressions; for the PLY parser also joint_view, inherit_linearly and the unnamed placeholder),
 * written against C++17 variadic templates. */
#pragma once
#include <type_traits>
namespace boost {
using std::is_same;
namespace mpl {
template <typename... T> struct vector { typedef vector type; };
template <typename A, typename B> struct pair { typedef A first; typedef B second; typedef pair type; };
struct _1 { }; struct _2 { };
struct _ { };                                     /* the unnamed placeholder: the argument of a unary lambda (ply_parser.hpp:146) */
template <typename Dummy = void> struct vector0 { typedef vector<> type; };
template <typename E> struct lambda { };

namespace detail {
    /* placeholder substitution */
    template <typename E, typename A, typename B> struct subst { typedef E type; };
    template <typename A, typename B> struct subst<_1, A, B> { typedef A type; };
    template <typename A, typename B> struct subst<_2, A, B> { typedef B type; };
    template <typename A, typename B> struct subst<_, A, B> { typedef A type; };
    template <template <typename...> class X, typename... Args, typename A, typename B>
    struct subst<X<Args...>, A, B> { typedef X<typename subst<Args, A, B>::type...> type; };
    /* a substituted expression is a metafunction call if it has a nested ::type, else a plain type */
    template <typename T, typename = void> struct eval { typedef T type; };
    template <typename T> struct eval<T, std::void_t<typename T::type>> { typedef typename T::type type; };
    template <typename Op> struct unwrap { typedef Op type; };
    template <typename E> struct unwrap<lambda<E>> { typedef E type; };
    template <typename Op, typename A, typename B> struct apply2 {
        typedef typename eval<typename subst<typename unwrap<Op>::type, A, B>::type>::type once;
        /* a lambda may evaluate to another lambda (ply_parser.hpp:146-158: pair_with<S>::type = pair<S, _>, which transform then applies to its
           elements): substitute and evaluate once more -- the identity on everything that holds no placeholder any more */
        typedef typename eval<typename subst<once, A, B>::type>::type type;
    };
    template <typename S1, typename S2> struct concat;
    template <typename... A, typename... B> struct concat<vector<A...>, vector<B...>> { typedef vector<A..., B...> type; };
}

template <typename Seq, typename T> struct push_back;
template <typename... S, typename T> struct push_back<vector<S...>, T> { typedef vector<S..., T> type; };

template <typename Seq, typename State, typename Op> struct fold;
template <typename State, typename Op> struct fold<vector<>, State, Op> { typedef State type; };
template <typename H, typename... R, typename State, typename Op> struct fold<vector<H, R...>, State, Op>
    : fold<vector<R...>, typename detail::apply2<Op, State, H>::type, Op> { };

template <typename Seq, typename Op> struct transform;
template <typename... S, typename Op> struct transform<vector<S...>, Op> { typedef vector<typename detail::apply2<Op, S, void>::type...> type; };

/* joint_view<A, B>: the concatenation of two sequences, each possibly still a metafunction call (its arguments arrive substituted, not evaluated) */
template <typename A, typename B> struct joint_view { typedef typename detail::concat<typename detail::eval<A>::type, typename detail::eval<B>::type>::type type; };

/* inherit_linearly<Seq, inherit<_1, F<_2>>>::type: one class that derives from F<T> for every T of Seq (the only form the PLY parser uses) */
template <typename A, typename B> struct inherit { };
template <typename Seq, typename Op> struct inherit_linearly;
template <typename... T, template <typename> class F> struct inherit_linearly<vector<T...>, inherit<_1, F<_2>>> { struct type : F<T>... { }; };
template <typename Seq, typename Op> struct inherit_linearly : inherit_linearly<typename detail::eval<Seq>::type, Op> { };

template <typename Seq> struct for_each_impl;
template <typename... S> struct for_each_impl<vector<S...>> { template <typename F> static void run(F f) { (f(S()), ...); } };
template <typename Seq, typename F> void for_each(F f) { for_each_impl<Seq>::run(f); }
} }
