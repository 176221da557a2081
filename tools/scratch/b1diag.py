import sys
sys.path.insert(0, ".")
from vits_amd import native
native.use_diagnostic_library()
sys.argv = ["b1.py", "fp32"]
exec(open("tools/scratch/b1.py").read())
