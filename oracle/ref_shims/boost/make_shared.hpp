// ORACLE -- TEST INFRASTRUCTURE ONLY.  boost::shared_ptr / make_shared as src/shapes/serialized.cpp:261-299 uses them: the standard library's.
#pragma once
#include <memory>
namespace boost { using std::shared_ptr; using std::make_shared; }
