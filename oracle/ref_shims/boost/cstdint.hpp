/* ref_shims: boost/cstdint.hpp as src/shapes/ply/ply.hpp uses it */
#pragma once
#include <cstdint>
namespace boost { using std::int8_t; using std::int16_t; using std::int32_t; using std::int64_t; using std::uint8_t; using std::uint16_t; using std::uint32_t; using std::uint64_t; }
