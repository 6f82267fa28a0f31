import sys, os
sys.path.insert(0, "/root/repo/tools/probes")
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("chk", "/root/repo/tools/probes/attn_bwd64_check.py")
src = open("/root/repo/tools/probes/attn_bwd64_check.py").read().split("for shp in [(1, 2, 64, 64)")[0]
exec(src)
for shp in [(1, 16, 1024, 1024), (2, 16, 1024, 1024), (4, 16, 1024, 1024), (8, 16, 1024, 1024), (16, 16, 1024, 1024), (4, 12, 1370, 1370), (8, 16, 196, 196), (64, 16, 196, 196)]:
    case(*shp, ref=False, time=True)
