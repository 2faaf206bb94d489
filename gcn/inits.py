"""Alias of gcn_amd.inits under the reference module name gcn.inits."""
import sys
from gcn_amd import inits as _impl
sys.modules[__name__] = _impl
