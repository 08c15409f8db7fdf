import sys; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
t,n=scenes.chap5_scene(); hdr,cache=scenes.hdr_and_cache()
sc=rt.Scene(t,n,hdr,cache,device=0)
eye,cam=rt.camera(0,20,4)
d=rt.make_desc(1920,1080,mode=5,eye=eye,camera_rotate=cam)
st=sc.render_stats(d,16,16); print(st)
