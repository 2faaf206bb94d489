"""Alias of gcn_amd.layers under the reference module name gcn.layers."""
import sys
from gcn_amd import layers as _impl
sys.modules[__name__] = _impl
