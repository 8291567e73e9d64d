"""Diagnostic (GPU): the saturating edge-MLP case gain=400, bias=0.5 at L=3 under pq_exp / fold / nt_min_wgs combinations --
prints the node-encoder gradient lines of the oracle comparison."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers
from oracle import sgnn_oracle as orc
import test_gpu_parity as tp
from drl_urban_planning_amd import native
DEV = 'cuda:0'

def tune(k, v):
    native.check(native.lib().upamd_tune(k.encode(), int(v)), 'tune')

def run(gain, bias, L, pq_exp, fold, minwgs):
    tune('pq_exp', pq_exp); tune('fold_layer1', fold); tune('nt_min_wgs', minwgs)
    D, heads, T, n_range = 64, 2, 6, (30, 60)
    cfg, sd, replay = tp._random_case(D, L, heads, (64, 16), (32, 1), (32, 1), (32, 32, 1), T, n_range[1] + 5,
                                      int(5.55 * n_range[1]) + 10, seed=33, road_fraction=0.3, n_range=n_range)
    sd = dict(sd)
    for k in list(sd):
        if 'edge_fc_layers' in k:
            sd[k] = sd[k] * gain if k.endswith('weight') else sd[k] + bias
    try:
        tp._check_against_oracle(cfg, sd, replay, heads, T, tol=3e-4)
        print('gain %g bias %g L %d pq_exp %d fold %d minwgs %d: OK' % (gain, bias, L, pq_exp, fold, minwgs))
    except AssertionError as e:
        msg = str(e)
        print('gain %g bias %g L %d pq_exp %d fold %d minwgs %d: FAIL' % (gain, bias, L, pq_exp, fold, minwgs))
        for line in msg.splitlines():
            if 'node_encoder' in line or 'edge_fc' in line or 'mismatch' in line or 'Mismatch' in line or 'Max abs' in line:
                print('   ', line[:200])

for L in (2, 3):
    for pq_exp, fold, minwgs in ((0, 1, 128), (0, 1, 1), (1, 1, 1), (1, 0, 1), (0, 0, 1)):
        run(400.0, 0.5, L, pq_exp, fold, minwgs)
