#pragma once
#include "../pangolin.h"
