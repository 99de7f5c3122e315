"""A/B helper: python tools/ab_bench.py <option-key> <value> [bench.py args...] -- sets a tss_set_option switch, then runs bench.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_semantic_segmentation_amd import _native as N  # noqa: E402

key, value = int(sys.argv[1]), int(sys.argv[2])
N.call('tss_set_option', key, value)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(root, 'bench.py')] + sys.argv[3:]
exec(compile(open(sys.argv[0]).read(), sys.argv[0], 'exec'))
