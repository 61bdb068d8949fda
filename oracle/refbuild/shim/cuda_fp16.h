#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
/* CUDA offers atomicAdd(__half2*, __half2); the HIP headers do not.  The reference only reaches it on its fp16 path
   (never exercised by the fp32 parity tests); a CAS loop keeps that path compilable and correct. */
__device__ inline __half2 atomicAdd(__half2* address, __half2 val) {
    unsigned int* p = reinterpret_cast<unsigned int*>(address);
    unsigned int old = *p, assumed;
    __half2 prev;
    do {
        assumed = old;
        prev = *reinterpret_cast<__half2*>(&assumed);
        __half2 sum = __hadd2(prev, val);
        old = atomicCAS(p, assumed, *reinterpret_cast<unsigned int*>(&sum));
    } while (old != assumed);
    return prev;
}
