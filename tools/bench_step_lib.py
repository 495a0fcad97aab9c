"""bench.py against another build of the library (SVDQ_LIB=path; tools only: the product loads nunchaku_amd/csrc/libsvdq_amd.so):
same-box A/B of the whole denoise step.  Arguments are bench.py's.  An older build is accepted when its argument structs are the
current ones (ABI 17 -> 18 only added profiler sub-classes): the binding's ABI check is pointed at the library's own number."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nunchaku_amd._lib as _L
_L._LIB_PATH = os.path.abspath(os.environ.get("SVDQ_LIB", _L._LIB_PATH))
_so = ctypes.CDLL(_L._LIB_PATH)
_v = _so.svdq_abi_version()
if _v in (17, 18):
    _L.ABI_VERSION = _v
for _name in list(_L.EXPORTS):  # entry points an older build does not have (host-side queries added later) are simply not bound
    if not hasattr(_so, _name):
        del _L.EXPORTS[_name]
import bench
bench.main()
