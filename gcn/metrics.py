"""Alias of gcn_amd.metrics under the reference module name gcn.metrics."""
import sys
from gcn_amd import metrics as _impl
sys.modules[__name__] = _impl
