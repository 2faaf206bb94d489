"""Drop-in alias: ``gcn.layers``, ``gcn.models``, ``gcn.utils``, ``gcn.metrics``
and ``gcn.inits`` are the MI355X implementations in ``gcn_amd`` under the
reference package's module names (reference gcn/__init__.py)."""
from __future__ import print_function
from __future__ import division
