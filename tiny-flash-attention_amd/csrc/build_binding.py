"""Compile csrc/attention_api.cpp into lib/attention_cutlass<EXT>.so (a torch extension module with the
reference's module and function names) with g++ — no setup.py, no cmake.  Links libtfa_hip.so by rpath."""
import os
import subprocess
import sys
import sysconfig

import pybind11
import torch
from torch.utils import cpp_extension as ce


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(here))
    lib = os.path.join(os.path.dirname(here), "lib")
    out = os.path.join(lib, "attention_cutlass" + sysconfig.get_config_var("EXT_SUFFIX"))
    src = os.path.join(here, "attention_api.cpp")
    deps = [src, os.path.join(root, "include", "tfa.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        print(f"up to date: {out}")
        return
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    incs = ce.include_paths() + [sysconfig.get_paths()["include"], pybind11.get_include(), "/opt/rocm/include",
                                 os.path.join(root, "include")]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=attention_cutlass",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{i}" for i in incs]
    cmd += [src, "-o", out, f"-L{tl}", f"-L{lib}", "-ltfa_hip", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
            "-ltorch_hip", "-ltorch_python", f"-Wl,-rpath,{tl}", "-Wl,-rpath,$ORIGIN"]
    print(" ".join(cmd))
    subprocess.check_call(cmd)


if __name__ == "__main__":
    sys.exit(main())
