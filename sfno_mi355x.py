"""File-path registration for the reference's model registry, no installation needed:

    nettype: "/path/to/repo/sfno_mi355x.py:SphericalFourierNeuralOperatorNet"

``makani/models/model_registry.py:63-79`` loads ``path:Name`` with ``spec_from_file_location`` -- a bare module outside
any package, so this shim puts the repository on ``sys.path`` and re-exports the network classes.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from makani_amd.sfnonet import FourierNeuralOperatorNet, SphericalFourierNeuralOperatorNet  # noqa: E402,F401
