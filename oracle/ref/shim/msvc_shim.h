// Forced-include compatibility shim: lets the *unmodified* reference BVH
// sources (MSVC dialect) compile with clang++ on Linux. Test infrastructure only.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <csignal>
#include <cmath>
#include <new>
#include <thread>
#include <mutex>
#include <condition_variable>
#define __debugbreak() (fflush(stdout), raise(SIGTRAP))
#define __forceinline __attribute__((always_inline))
typedef int errno_t;
static inline errno_t fopen_s(FILE ** f, const char * name, const char * mode) { *f = fopen(name, mode); return *f ? 0 : errno; }
static inline size_t fread_s(void * buf, size_t, size_t es, size_t n, FILE * f) { return fread(buf, es, n, f); }
template<size_t N> static inline int strerror_s(char (&buf)[N], int e) { strncpy(buf, strerror(e), N); return 0; }
#define fprintf_s fprintf
static inline void * _aligned_malloc(size_t size, size_t align) { void * p = nullptr; return posix_memalign(&p, align, size) == 0 ? p : nullptr; }
static inline void _aligned_free(void * p) { free(p); }
using std::isinf; using std::isnan;
