#pragma once
#include <unordered_map>
namespace boost { template <typename... T> using unordered_map = std::unordered_map<T...>; }
