#pragma once
#include <boost/filesystem.hpp>
