#pragma once
#include "rtcore.h"
