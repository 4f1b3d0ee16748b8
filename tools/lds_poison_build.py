#!/usr/bin/env python3
"""Debug build of libnerfacc_hip.so whose kernels start by filling their workgroup's WHOLE LDS allocation with a pattern
(0xAAAAAAAA on even workgroups, 0x55555555 on odd ones): a kernel that reads an LDS word it has not written this launch gives
the same results as always in a process of its own (the word holds what its own earlier workgroups left there) and garbage when
other processes' kernels ran on the CU in between — the poison makes such a read show in the ordinary single-process tests.

  python tools/lds_poison_build.py        ->  build/poison/nerfacc_amd/libnerfacc_hip.so   (the tree's sources are not touched)
"""
import os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "build", "poison")

POISON = r'''
#ifndef NFA_POISON_DEFINED
#define NFA_POISON_DEFINED
__device__ __attribute__((noinline)) static void nfa_lds_poison() {
    const unsigned bytes = ((const __attribute__((address_space(4))) unsigned *)__builtin_amdgcn_dispatch_ptr())[7];   // group_segment_size
    volatile __attribute__((address_space(3))) unsigned *p = (volatile __attribute__((address_space(3))) unsigned *)0;
    const unsigned pat = (blockIdx.x & 1) ? 0x55555555u : 0xAAAAAAAAu;
    for (unsigned i = threadIdx.x; i < bytes / 4; i += blockDim.x) p[i] = pat;
    __syncthreads();
}
#endif
'''

def main():
    shutil.rmtree(DST, ignore_errors=True)
    os.makedirs(os.path.join(DST, "nerfacc_amd"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(DST, "include"))
    src = os.path.join(ROOT, "nerfacc_amd", "csrc")
    dst = os.path.join(DST, "nerfacc_amd", "csrc")
    os.makedirs(dst)
    n_kernels = 0
    for f in sorted(os.listdir(src)):
        if not f.endswith((".hip", ".hpp", ".cpp")) and f != "Makefile":
            continue
        text = open(os.path.join(src, f)).read()
        if f.endswith((".hip", ".hpp")) and "__global__" in text:
            out, pos = [], 0
            for m in re.finditer(r"__global__", text):
                # the kernel's body: the first '{' behind the parameter list's closing parenthesis
                i = text.index("(", m.end())
                # (__launch_bounds__(...) comes first: skip balanced groups until the one followed by '{')
                while True:
                    depth, j = 0, i
                    while True:
                        c = text[j]
                        depth += c == "("
                        depth -= c == ")"
                        j += 1
                        if depth == 0:
                            break
                    k = j
                    while text[k] in " \t\r\n":
                        k += 1
                    if text[k] == "{":
                        break
                    i = text.index("(", j)
                out.append(text[pos:k + 1] + " nfa_lds_poison(); ")
                pos = k + 1
                n_kernels += 1
            out.append(text[pos:])
            text = "".join(out)
            # the helper goes behind common.hpp's include (it needs nothing from it, but hip_runtime.h has to be there)
            text = text.replace('#include "common.hpp"', '#include "common.hpp"' + POISON, 1)
        open(os.path.join(dst, f), "w").write(text)
    print("kernels poisoned:", n_kernels)
    subprocess.check_call(["make", "-C", dst, "-j4"])
    print(os.path.join(DST, "nerfacc_amd", "libnerfacc_hip.so"))

if __name__ == "__main__":
    sys.exit(main())
