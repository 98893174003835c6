"""pip install -e .   (builds libq1env.so in-tree with hipcc for gfx950, then installs the two pure-Python packages:
q1physrl_amd and the drop-in namespace q1physrl_env)."""
import os
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        sys.path.insert(0, ROOT)
        from q1physrl_amd import build as hipbuild
        print("building", hipbuild.build_lib(verbose=True))
        super().run()


setup(
    name="q1physrl_amd",
    version="0.1.0",
    description="Quake-1 player-movement RL environment of matthewearl/q1physrl as MI355X (gfx950) HIP kernels behind the reference's env API",
    packages=["q1physrl_amd", "q1physrl_env"],
    package_data={"q1physrl_amd": ["libq1env.so", "csrc/*", ]},
    data_files=[("include", ["include/q1env.h"])],
    install_requires=["numpy"],
    extras_require={"torch": ["torch"]},
    cmdclass={"build_py": BuildWithHip},
    python_requires=">=3.9",
)
