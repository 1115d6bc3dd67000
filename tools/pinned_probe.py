"""Developer probe: can a kernel store straight into torch pinned host memory (same virtual address)?"""
import ctypes, os, subprocess, sys, tempfile, torch
src = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void poke(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
extern "C" int run(int* host_ptr, int v, void* stream) {
    hipLaunchKernelGGL(poke, dim3(1), dim3(1), 0, (hipStream_t)stream, host_ptr, v);
    return (int)hipGetLastError();
}
'''
d = tempfile.mkdtemp()
open(d + "/p.hip", "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", d + "/p.hip", "-o", d + "/p.so"])
lib = ctypes.CDLL(d + "/p.so")
lib.run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
t = torch.zeros((8,), dtype=torch.int32).pin_memory()
x = torch.zeros(1, device="cuda")
rc = lib.run(t.data_ptr() + 4 * 3, 4242, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("rc", rc, "pinned after kernel:", t.tolist())
