// shim_pre.h -- TEST INFRASTRUCTURE (build container only).
//
// Lets g++ compile, as ordinary host C++, the anonymous-namespace kernel bodies
// of /root/reference/pnpmodules/neural_renderer/neural_renderer/cuda/
// rasterize_cuda_kernel.cu (lines 22-593) exactly where they lie: the Makefile
// pipes those lines between this header and shim_post.inc straight into g++;
// no reference text is written anywhere in this repository.
//
// What this header supplies is the CUDA *execution-model vocabulary* the bodies
// use, not an emulation of any CUDA library: the __global__/__device__
// qualifiers, the per-thread index variables, CUDA's mixed float/double
// min/max overload set and a serial atomicAdd.  Conversions float->int in the
// bodies follow x86 semantics here (CUDA saturates); they only differ for
// non-finite edge crossings, which the fixtures avoid ("Oracle").
#pragma once
#include <cstdint>
#include <cstddef>
#include <cmath>
#include <cstdio>
#include <vector>

#define __global__
#define __device__

struct d3m_dim3_ { int x, y, z; };
static thread_local d3m_dim3_ blockIdx = {0, 0, 0}, blockDim = {1, 1, 1}, threadIdx = {0, 0, 0};

using std::ceil;
using std::floor;

// CUDA's fmax/fmin semantics: a NaN operand is dropped.
static inline double max(float a, double b) { return std::fmax((double)a, b); }
static inline double max(double a, float b) { return std::fmax(a, (double)b); }
static inline double max(double a, double b) { return std::fmax(a, b); }
static inline float max(float a, float b) { return std::fmax(a, b); }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline double min(float a, double b) { return std::fmin((double)a, b); }
static inline double min(double a, float b) { return std::fmin(a, (double)b); }
static inline double min(double a, double b) { return std::fmin(a, b); }
static inline float min(float a, float b) { return std::fmin(a, b); }
static inline int min(int a, int b) { return a < b ? a : b; }

template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p += v; return o; }
