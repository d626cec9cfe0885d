/* tests/helpers/hipcount.c -- TEST INFRASTRUCTURE: an LD_PRELOAD interposer that counts the HIP runtime's allocation entry points.
 * tests/test_gpu_configs.py::test_nothing_is_allocated_inside_forward_and_decode runs a child process under it and asserts that the first
 * flm_forward / flm_decode_greedy of a context make no such call (the reference's "zero allocations during inference", transformer.cpp:110-130).
 * Built by __graft_entry__.build() into tests/helpers/libhipcount.so (gcc -shared -fPIC ... -ldl).  Never loaded by the product. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>

static long n_alloc = 0, n_free = 0;
static void* next(const char* name) { return dlsym(RTLD_NEXT, name); }

#define COUNTED(name, params, args)                                            \
    int name params {                                                          \
        static int (*real) params;                                             \
        if (!real) real = (int (*) params) next(#name);                        \
        __sync_fetch_and_add(&n_alloc, 1);                                     \
        return real ? real args : 2 /* hipErrorOutOfMemory */;                 \
    }
COUNTED(hipMalloc, (void** p, size_t n), (p, n))
COUNTED(hipExtMallocWithFlags, (void** p, size_t n, unsigned f), (p, n, f))
COUNTED(hipHostMalloc, (void** p, size_t n, unsigned f), (p, n, f))
COUNTED(hipMallocManaged, (void** p, size_t n, unsigned f), (p, n, f))
COUNTED(hipMallocAsync, (void** p, size_t n, void* s), (p, n, s))
COUNTED(hipMallocPitch, (void** p, size_t* pitch, size_t w, size_t h), (p, pitch, w, h))

int hipFree(void* p) {
    static int (*real)(void*);
    if (!real) real = (int (*)(void*)) next("hipFree");
    __sync_fetch_and_add(&n_free, 1);
    return real ? real(p) : 1;
}
long hipcount_allocs(void) { return n_alloc; }
long hipcount_frees(void) { return n_free; }
