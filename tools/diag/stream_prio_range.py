import ctypes as C, torch
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(), C.c_int()
print("rc", hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), "least", lo.value, "greatest", hi.value)
