#pragma once
#include <ATen/hip/HIPContext.h>
