"""Compile the reference's CPU attention (flash_attention_c/csrc/attn.cpp + ops.cu) UNMODIFIED
into oracle/_ref/_kernels<EXT>.so.  Test infrastructure only (see oracle/Makefile).
Recipe from SURVEY.md Appendix B: four empty CUDA shim headers first on the include path."""
import os
import subprocess
import sys
import sysconfig

import pybind11
import torch
from torch.utils import cpp_extension as ce


def main():
    refsrc, out = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    incs = ce.include_paths() + [sysconfig.get_paths()["include"], pybind11.get_include()]
    cmd = [
        "g++", "-O3", "-std=c++17", "-fopenmp", "-fPIC", "-shared", "-w",
        "-DTORCH_EXTENSION_NAME=_kernels",
        f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
        f"-I{os.path.join(here, 'shim')}",
    ] + [f"-I{i}" for i in incs] + [
        f"-I{refsrc}", os.path.join(refsrc, "attn.cpp"), "-x", "c++", os.path.join(refsrc, "ops.cu"),
        "-o", out, f"-L{tl}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", f"-Wl,-rpath,{tl}",
    ]
    print(" ".join(cmd))
    subprocess.check_call(cmd)


if __name__ == "__main__":
    main()
