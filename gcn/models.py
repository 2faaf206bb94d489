"""Alias of gcn_amd.models under the reference module name gcn.models."""
import sys
from gcn_amd import models as _impl
sys.modules[__name__] = _impl
