// Forwarding header: DynSLAM includes "../InfiniTAM/InfiniTAM/ITMLib/Engine/ITMMainEngine.h"
// (InfiniTamDriver.h:13).  With `src/InfiniTAM` pointing at this directory tree (a symlink, or
// `-I<repo>/shim/DynSLAM` so that the relative path resolves here) the host gets the dsr-backed
// ITMLib names of shim/ITMLib.h instead of the CUDA engines.
#pragma once
#include <iostream>
#include <string>

#include "../../../../ITMLib.h"

// upstream's ITMLib headers leak `using namespace std` into every includer and the host relies on it
// (unqualified `string`, `cout`, `endl`, `cerr`, `runtime_error`: InfiniTamDriver.h:46,220;
// InfiniTamDriver.cpp:30,172)
using namespace std;
