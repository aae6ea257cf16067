import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rays1bench_amd as r1
from rays1bench_amd import binding
w,h,spp=1200,800,10
sc=r1.create_large_scene(w,h); rend=r1.Renderer(0); rend.set_scene(sc)
for v in (4,5,5):
    img,rays,secs=rend.render(r1.make_params(w,h,spp,10001,variant=v)); print(v, rend.last_timing())
st=rend.last_stats(); info=rend.launch_info(); print(info); print(json.dumps(st,indent=1))
waves=info["blocks"]*4
print("mean wave cycles", st["cycles_wave"]/waves, "longest", st["longest_wave_cycles"], "shortest", st["shortest_wave_cycles"], "span", st["span_cycles"])
