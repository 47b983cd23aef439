"""Builds the Cython binding in-tree:  python suffixarray_amd/setup_cython.py build_ext --inplace
(needs libsa_hip.so from `python -m suffixarray_amd.build` first)."""
import os

import numpy
from Cython.Build import cythonize
from setuptools import Extension, setup

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.chdir(ROOT)

ext = Extension(
    "suffixarray_amd.suffix_array",
    sources=["suffixarray_amd/suffix_array.pyx"],
    include_dirs=[os.path.join(ROOT, "include"), numpy.get_include()],
    libraries=["sa_hip"],
    library_dirs=[HERE],
    runtime_library_dirs=["$ORIGIN"],
    extra_compile_args=["-O2"],
)
setup(name="suffixarray_amd_cython", ext_modules=cythonize([ext], language_level=3), script_args=None)
