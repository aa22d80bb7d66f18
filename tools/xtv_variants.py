import sys; sys.path.insert(0, '.')
from bess_amd import capi
names = {0:'U8 CG16 nt',1:'U8 CG16 plain',2:'U8 CG8 nt',3:'U8 CG8 plain',4:'U4 CG16 nt',5:'U4 CG16 plain',6:'U4 CG8 nt',7:'U2 CG16 nt',8:'U8 CG4 nt',9:'U4 CG4 nt'}
for rep in range(2):
    for v in range(10):
        g, ms = capi.op_xtv_bench(50000, 10000, v, 30)
        print("variant %d %-14s %8.1f GB/s  %.4f ms" % (v, names[v], g, ms))
print("copy", capi.op_stream_copy_gbps(1<<31, 10))
