"""Alias of gcn_amd.utils under the reference module name gcn.utils."""
import sys
from gcn_amd import utils as _impl
sys.modules[__name__] = _impl
