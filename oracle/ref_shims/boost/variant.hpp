#pragma once
#include <variant>
namespace boost {
template <typename... T> using variant = std::variant<T...>;
template <typename R> struct static_visitor { typedef R result_type; };
template <typename T, typename... A> T *get(std::variant<A...> *v) { return std::get_if<T>(v); }
template <typename T, typename... A> const T *get(const std::variant<A...> *v) { return std::get_if<T>(v); }
template <typename V, typename... A> typename V::result_type apply_visitor(const V &vis, const std::variant<A...> &v) {
    return std::visit([&](const auto &x) -> typename V::result_type { return vis(x); }, v);
}
template <typename V, typename... A> typename V::result_type apply_visitor(V &vis, const std::variant<A...> &v) {
    return std::visit([&](const auto &x) -> typename V::result_type { return vis(x); }, v);
}
}
