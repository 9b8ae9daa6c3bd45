#pragma once
#include <unordered_set>
namespace boost { template <typename... T> using unordered_set = std::unordered_set<T...>; }
