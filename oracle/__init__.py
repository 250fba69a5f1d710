"""CPU oracle for the OD-WSCL hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker / timed CPU baseline.  Nothing in
od_wscl_amd/ imports it; the product path has no CPU fallback.
"""
