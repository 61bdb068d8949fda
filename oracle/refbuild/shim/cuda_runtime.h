#pragma once
#include <hip/hip_runtime.h>
