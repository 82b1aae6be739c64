import ctypes as C, torch
torch.cuda.init()
h = C.CDLL("libamdhip64.so")
for name, val in (("MaxSharedMemoryPerBlock", None), ("SharedMemPerBlockOptin", None)):
    pass
import re
hdr = open("/opt/rocm/include/hip/hip_runtime_api.h").read()
# enum values: parse order in hipDeviceAttribute_t
m = re.search(r"typedef enum hipDeviceAttribute_t \{(.*?)\} hipDeviceAttribute_t;", hdr, re.S)
names = [l.split(",")[0].split("=")[0].strip() for l in m.group(1).split("\n") if l.strip().startswith("hipDeviceAttribute")]
vals = {}
cur = 0
for l in m.group(1).split("\n"):
    l = l.strip()
    if not l.startswith("hipDeviceAttribute"): continue
    nm = l.split(",")[0]
    if "=" in nm:
        n, v = nm.split("=")
        try: cur = int(v.strip(), 0)
        except Exception: cur = vals.get(v.strip(), cur)
        nm = n.strip()
    vals[nm.strip()] = cur
    cur += 1
for k in ("hipDeviceAttributeMaxSharedMemoryPerBlock", "hipDeviceAttributeSharedMemPerBlockOptin", "hipDeviceAttributeMaxSharedMemoryPerMultiprocessor"):
    v = C.c_int(0)
    r = h.hipDeviceGetAttribute(C.byref(v), vals[k], 0)
    print(k, vals[k], "rc", r, "value", v.value)
