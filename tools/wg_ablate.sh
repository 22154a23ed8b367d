# usage (GPU box): tools/wg_ablate.sh  -- the head weight-gradient launches under ED3DGS_WG_ABLATE (results are then WRONG: timing only).
# (ED3DGS_WGRAD_SEPARATE=1: the two kinds as two launches, so that each is timed on its own; 16 = no final adds, 32 = no dW3, 64 = no g_y . W3.)
# 8 = "3 compute waves + 1 loader wave" emulated (csrc/deform.hip HeadWgradArgs.ablate): time x 4/3 against ablate 0 prices that design.
for a in ${WG_LIST:-0 8 0 8 1 2 4}; do ED3DGS_WGRAD_SEPARATE=1 ED3DGS_WG_ABLATE=$a python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernels']; n=k['deform_head_wgrad_tr_kernel<false>']['avg_launch_ms']; w=k['deform_head_wgrad_tr_kernel<true>']['avg_launch_ms']
print('ablate $a', 'step %.3f'%d['ms_per_step'], 'narrow %.4f (x4/3 = %.4f)'%(n, n*4/3), 'wide %.4f (x4/3 = %.4f)'%(w, w*4/3))"; done
