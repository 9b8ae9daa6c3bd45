#pragma once
#include <tuple>
namespace boost {
template <typename... T> using tuple = std::tuple<T...>;
using std::make_tuple; using std::tie;
template <size_t I, typename... T> auto &get(std::tuple<T...> &t) { return std::get<I>(t); }
template <size_t I, typename... T> const auto &get(const std::tuple<T...> &t) { return std::get<I>(t); }
}
