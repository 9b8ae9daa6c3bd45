#pragma once
#include <functional>
namespace boost { using std::bind; using std::ref; using std::cref; }
using namespace std::placeholders;
