/* Build shim for oracle/_ref only: lets hipcc read the reference's CUDA sources where they lie (test infrastructure). */
#pragma once
#include <hip/hip_runtime.h>
