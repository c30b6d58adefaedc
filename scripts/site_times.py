"""Per kernel-site durations (HIP events around every launch of one tagged site, weight-gradient stream serialised),
one site at a time: the table behind DESIGN.md section 6.   python scripts/site_times.py [--frames F] [--precision P]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd')); sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hipvae import Engine  # noqa: E402
from hipvae.dp import Stepper  # noqa: E402
TAGS = ('prep enc0_fwd enc1_split enc1_fwd stats_enc1 enc2_split enc2_fwd stats_enc2 enc3_split enc3_fwd stats_enc3 enc4_split enc4_fwd stats_enc4 heads_split heads_fwd '
        'reparam merge_split merge_fwd dec0_split dec0_fwd stats_dec0 dec1_split dec1_fwd stats_dec1 dec2_split dec2_fwd dec2_stats_planes dec3_fwd loss '
        'dxh_post dec3_wgrad dec3_row512 dec3_bias dec3_dgrad lnb_dec2 dec2_bwd dec2_gsplit dec2_asplit dec2_wgrad dec2_dgrad lnb_dec1 dec1_bwd dec1_gsplit dec1_asplit '
        'dec1_wgrad dec1_dgrad lnb_dec0 dec0_gsplit dec0_asplit dec0_wgrad dec0_dgrad merge_dsplit merge_wgrad merge_segsum merge_small merge_dgrad reparam_bwd '
        'heads_dsplit heads_wgrad heads_dgrad lnb_enc4 enc4_dsplit enc4_wgrad enc4_dgrad lnb_enc3 enc3_gsplit enc3_asplit enc3_wgrad enc3_dgrad lnb_enc2 '
        'enc2_gsplit enc2_asplit enc2_wgrad enc2_dgrad lnb_enc1 enc1_bwd enc1_gsplit enc1_asplit enc1_wgrad enc1_dgrad lnb_enc0 enc0_wgrad enc0_bwd enc0_reduce').split()
ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=32768)
ap.add_argument('--precision', default='auto')
ap.add_argument('--steps', type=int, default=4)
ap.add_argument('--tags', default='', help='comma separated subset of the site tags')
a = ap.parse_args()
arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
eng = Engine(arch, precision=a.precision)
eng.init_params(0)
eng.set_tuned_masks(0xffffffff, 0xbfffffff)
st = Stepper(eng, 1e-4, 0.5, 0.999)
g = torch.Generator().manual_seed(1)
x = (torch.rand(a.frames, 513, generator=g) * 2 - 1).cuda()
y = torch.randint(0, 10, (a.frames,), generator=g).cuda()
for _ in range(2):
    st.step(x, y)
tot = 0.0
for tag in (a.tags.split(',') if a.tags else TAGS):
    eng.timer_select(tag)
    for _ in range(a.steps):
        st.step(x, y)
    torch.cuda.synchronize()
    ms, n = eng.timer_read()
    if n:
        tot += ms / a.steps
        print('%-14s %8.1f us  (%d launches/step)' % (tag, 1e3 * ms / n, n // a.steps))
eng.timer_select(None)
print('sum of tagged sites per step: %.3f ms' % tot)
