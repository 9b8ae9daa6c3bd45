/* ref_shims: the sliver of Boost.MPL that src/libcore/fmtconv.cpp:1171-1208 uses (type lists, fold / transform /
 * for_each with placeholder lambda expressions), written against C++17 variadic templates. */
#pragma once
#include <type_traits>
namespace boost {
using std::is_same;
namespace mpl {
template <typename... T> struct vector { typedef vector type; };
template <typename A, typename B> struct pair { typedef A first; typedef B second; };
struct _1 { }; struct _2 { };
template <typename E> struct lambda { };

namespace detail {
    /* placeholder substitution */
    template <typename E, typename A, typename B> struct subst { typedef E type; };
    template <typename A, typename B> struct subst<_1, A, B> { typedef A type; };
    template <typename A, typename B> struct subst<_2, A, B> { typedef B type; };
    template <template <typename...> class X, typename... Args, typename A, typename B>
    struct subst<X<Args...>, A, B> { typedef X<typename subst<Args, A, B>::type...> type; };
    /* a substituted expression is a metafunction call if it has a nested ::type, else a plain type */
    template <typename T, typename = void> struct eval { typedef T type; };
    template <typename T> struct eval<T, std::void_t<typename T::type>> { typedef typename T::type type; };
    template <typename Op> struct unwrap { typedef Op type; };
    template <typename E> struct unwrap<lambda<E>> { typedef E type; };
    template <typename Op, typename A, typename B> struct apply2 {
        typedef typename eval<typename subst<typename unwrap<Op>::type, A, B>::type>::type type;
    };
}

template <typename Seq, typename T> struct push_back;
template <typename... S, typename T> struct push_back<vector<S...>, T> { typedef vector<S..., T> type; };

template <typename Seq, typename State, typename Op> struct fold;
template <typename State, typename Op> struct fold<vector<>, State, Op> { typedef State type; };
template <typename H, typename... R, typename State, typename Op> struct fold<vector<H, R...>, State, Op>
    : fold<vector<R...>, typename detail::apply2<Op, State, H>::type, Op> { };

template <typename Seq, typename Op> struct transform;
template <typename... S, typename Op> struct transform<vector<S...>, Op> { typedef vector<typename detail::apply2<Op, S, void>::type...> type; };

template <typename Seq> struct for_each_impl;
template <typename... S> struct for_each_impl<vector<S...>> { template <typename F> static void run(F f) { (f(S()), ...); } };
template <typename Seq, typename F> void for_each(F f) { for_each_impl<Seq>::run(f); }
} }
