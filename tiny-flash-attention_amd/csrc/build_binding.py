"""Compile csrc/attention_api.cpp into the three torch extension modules that carry the reference's module and function
names — lib/attention_cutlass<EXT>.so, lib/attention_cuda<EXT>.so, lib/_kernels<EXT>.so — with g++ (no setup.py, no
cmake).  Each links libtfa_hip.so by rpath.  Reference build boundary: flash_attention_cutlass/build.py:42-84."""
import os
import subprocess
import sys
import sysconfig

import pybind11
import torch
from torch.utils import cpp_extension as ce

# module name -> TFA_BINDING selector in attention_api.cpp
MODULES = {"attention_cutlass": 1, "attention_cuda": 2, "_kernels": 3}


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(here))
    lib = os.path.join(os.path.dirname(here), "lib")
    src = os.path.join(here, "attention_api.cpp")
    deps = [src, os.path.join(root, "include", "tfa.h"), os.path.abspath(__file__)]
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    incs = ce.include_paths() + [sysconfig.get_paths()["include"], pybind11.get_include(), "/opt/rocm/include",
                                 os.path.join(root, "include")]
    procs = []
    for name, sel in MODULES.items():
        out = os.path.join(lib, name + sysconfig.get_config_var("EXT_SUFFIX"))
        if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
            print(f"up to date: {out}")
            continue
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
               f"-DTORCH_EXTENSION_NAME={name}", f"-DTFA_BINDING={sel}",
               f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
        cmd += [f"-I{i}" for i in incs]
        cmd += [src, "-o", out, f"-L{tl}", f"-L{lib}", "-ltfa_hip", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
                "-ltorch_hip", "-ltorch_python", f"-Wl,-rpath,{tl}", "-Wl,-rpath,$ORIGIN"]
        print(" ".join(cmd))
        procs.append((name, subprocess.Popen(cmd)))
    rc = 0
    for name, p in procs:
        if p.wait() != 0:
            print(f"build of {name} failed", file=sys.stderr)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
