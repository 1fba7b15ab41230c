"""`pip install -e .` for users of the reference's `python setup.py install` (setup.py there builds the Cython kernels):
here the one native artefact is cytvdn_amd/libtvdn_hip.so, built IN TREE for gfx950 by `make -C cytvdn_amd/csrc` (hipcc).
The package has no CPU fallback; importing it works anywhere, calling it needs an MI355X."""
import os
import subprocess

from setuptools import Command, setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildHip(Command):
    description = "compile the HIP kernels and the C ABI into cytvdn_amd/libtvdn_hip.so"
    user_options = []

    def initialize_options(self):
        pass

    def finalize_options(self):
        pass

    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cytvdn_amd", "csrc")])


class BuildPy(build_py):
    def run(self):
        self.run_command("build_hip")
        super().run()


setup(
    name="cytvdn_amd",
    version="0.6.0",
    description="MI355X-native anisotropic TV denoising behind the cyTVDN API (denoise3D / denoise4D)",
    packages=["cytvdn_amd"],
    package_data={"cytvdn_amd": ["libtvdn_hip.so"]},
    python_requires=">=3.9",
    install_requires=["numpy"],          # torch (ROCm build) is the HBM allocator: installed from AMD's index, not from PyPI
    cmdclass={"build_hip": BuildHip, "build_py": BuildPy},
)
