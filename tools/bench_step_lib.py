"""bench.py against another build of the library (SVDQ_LIB=path; tools only: the product loads nunchaku_amd/csrc/libsvdq_amd.so):
same-box A/B of the whole denoise step.  Arguments are bench.py's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nunchaku_amd._lib as _L
_L._LIB_PATH = os.path.abspath(os.environ.get("SVDQ_LIB", _L._LIB_PATH))
import bench
bench.main()
