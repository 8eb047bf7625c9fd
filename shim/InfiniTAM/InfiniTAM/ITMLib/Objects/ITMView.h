// Forwarding header (InstRecLib/InstanceView.h:4): see ../Engine/ITMMainEngine.h
#pragma once
#include "../Engine/ITMMainEngine.h"
