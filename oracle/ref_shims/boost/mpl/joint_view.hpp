#pragma once
#include <boost/mpl/vector.hpp>
