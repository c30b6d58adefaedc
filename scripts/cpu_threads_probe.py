import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'vae-npvc_amd'))
import bench
arch = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
for th in (8, 16, 32, 64):
    os.environ['VAENPVC_CPU_THREADS'] = str(th)
    t = time.time(); r = bench.cpu_baseline(arch, 3.0)
    print('threads', th, 'frames/s %.1f' % r['value'], 'ms/step %.1f' % r['ms_per_step'], 'wall %.1f' % (time.time() - t), flush=True)
