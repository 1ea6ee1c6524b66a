"""tools/size_scan.py under a kernel-selection variant (pffft_hip_set_variant), e.g. 83 = the streaming passes instead of the
odd-stage tile plans:   python tools/scan_variant.py 83 10240 600000 f32"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import pffft_amd as pa
v = int(sys.argv[1]); sys.argv = [sys.argv[0]] + sys.argv[2:]
pa.set_variant(v)
import size_scan
sys.exit(size_scan.main())
