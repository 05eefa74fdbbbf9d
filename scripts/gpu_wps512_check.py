import sys,os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as R
from oracle import oracle as O
import parity_cases as pc
lib=N.NativeLib(sys.argv[1])
for shape in ((300,40,120,8),(40,5,300,8),(200,100,200,8)):
    n,ne,ni,B=shape
    try:
        x,y,z,info=pc.case_random_batch(lib,O,R,n,ne,ni,B=B)
        print(os.path.basename(sys.argv[1]),shape,'ok', 'nan' if np.isnan(x).any() else '')
    except AssertionError as e:
        print(os.path.basename(sys.argv[1]),shape,'FAIL',str(e)[:100])
