#pragma once
#include "../opencv.hpp"
