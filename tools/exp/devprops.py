import ctypes, torch
p = torch.cuda.get_device_properties(0)
print(p)
for a in dir(p):
    if not a.startswith('_'):
        try: print(a, getattr(p, a))
        except Exception as e: pass
hip = ctypes.CDLL("libamdhip64.so")
val = ctypes.c_int()
# hipDeviceAttribute_t values (hip_runtime_api.h): look up a few by number
names = {}
import re
try:
    txt = open('/opt/rocm/include/hip/hip_runtime_api.h').read()
    m = re.search(r'typedef enum hipDeviceAttribute_t \{(.*?)\} hipDeviceAttribute_t;', txt, re.S)
    body = m.group(1)
    idx = 0
    for line in body.split('\n'):
        line = line.split('//')[0].strip().rstrip(',')
        if not line or line.startswith('/*') or line.startswith('*'): continue
        mm = re.match(r'(hipDeviceAttribute\w+)\s*(=\s*(\w+))?', line)
        if not mm: continue
        if mm.group(3):
            try: idx = int(mm.group(3), 0)
            except: idx = names.get(mm.group(3), idx)
        names[mm.group(1)] = idx
        idx += 1
except Exception as e:
    print("parse fail", e)
for k, v in names.items():
    if any(s in k for s in ("Shared", "Lds", "LDS", "MultiProcessor", "Regs", "Clock", "L2", "Wave", "Compute")):
        r = hip.hipDeviceGetAttribute(ctypes.byref(val), v, 0)
        print(k, v, val.value if r == 0 else "err%d" % r)
