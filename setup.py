"""Build hook of `pip install [-e] .`: the counterpart of the reference's CUDAExtension recipe
(Edit_core/thirdparties/diff-gaussian-rasterization/setup.py:17-34), without nvcc, hipify or torch's extension builder.
`pip install [-e] .` provides `diff_gaussian_rasterization` and `simple_knn` (the two import names the reference's README installs,
/root/reference/README.md:24-28) and `youreditableavatar_amd`; the build step compiles the native libraries in-tree first."""
import os
import sys

from setuptools import setup
from setuptools.dist import Distribution
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

HERE = os.path.dirname(os.path.abspath(__file__))


def _build_native():
    sys.path.insert(0, HERE)
    from youreditableavatar_amd import build as b
    lib = b.build_native(verbose=True)             # hipcc --offload-arch=gfx950 -> youreditableavatar_amd/lib/libtgs_raster.so
    ext = b.build_torch_ext(verbose=True)          # g++ + pybind11 -> youreditableavatar_amd/diff_gaussian_rasterization/_Cext*.so
    print("native:", lib, ext)


class BinaryDistribution(Distribution):
    """the package ships compiled code (libtgs_raster.so, _Cext*.so): the wheel is tagged for this interpreter and platform"""

    def has_ext_modules(self):
        return True


class BuildPy(build_py):
    def run(self):
        _build_native()                            # before the package data is collected, so the .so files travel in the wheel
        super().run()


class Develop(develop):
    def run(self):
        _build_native()
        super().run()


setup(
    name="youreditableavatar-amd",
    version="0.4.0",
    description="MI355X-native (gfx950) differentiable 3D Gaussian rasterizer: drop-in for diff_gaussian_rasterization and simple_knn of liuhx02/YourEditableAvatar",
    python_requires=">=3.9",
    install_requires=["torch", "numpy"],
    packages=["youreditableavatar_amd", "youreditableavatar_amd.diff_gaussian_rasterization", "youreditableavatar_amd.simple_knn",
              "diff_gaussian_rasterization", "simple_knn"],
    package_data={"youreditableavatar_amd": ["lib/libtgs_raster.so", "lib/.source_hash"],
                  "youreditableavatar_amd.diff_gaussian_rasterization": ["_Cext*.so", ".source_hash"]},
    include_package_data=False,
    zip_safe=False,
    cmdclass={"build_py": BuildPy, "develop": Develop},
    distclass=BinaryDistribution,
)
