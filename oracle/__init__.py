"""TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's arithmetic
(see gcn_oracle.c for the file:line map).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; nothing under
gcn_amd/ does."""
from .cpu import *  # noqa: F401,F403
