// Forwarding header (Input.h:10): see ../Engine/ITMMainEngine.h
#pragma once
#include "../Engine/ITMMainEngine.h"
