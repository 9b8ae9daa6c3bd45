#pragma once
#include <boost/bimap.hpp>
