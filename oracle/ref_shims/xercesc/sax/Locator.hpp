/* ORACLE -- TEST INFRASTRUCTURE ONLY: see xercesc/_mini.hpp */
#pragma once
#include <xercesc/_mini.hpp>
