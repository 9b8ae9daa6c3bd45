#pragma once
#include <functional>
namespace boost { template <typename S> using function = std::function<S>; }
