import sys; sys.path.insert(0,'.')
import numpy as np
from sfm_amd import matcher
from sfm_amd._lib import ptr
ctx=matcher.Context(0)
m=np.zeros(1,np.int32)
ctx.check(ctx._L.sfm_gpu_debug_sqrt_probe(ctx.handle, ptr(m)))
print("sqrtf mismatches over 2^24 integers:", m[0])
