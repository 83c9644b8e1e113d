import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, oracle, helpers
from pointcloudpdf_amd import _native
from test_pointops2_cpu import window_graph
be, ob = _native.hip_backend(), oracle.backend()
for (h, d, L, deg, n) in [(3, 16, 48, 40, 300), (3, 16, 64, 40, 300), (3, 16, 64, 90, 5500), (6, 16, 64, 60, 700), (12, 16, 64, 60, 400), (24, 16, 64, 30, 300)]:
    G = window_graph(5, n, h, d, L, deg)
    D = lambda t: t.cuda()
    a_o = ob.attention_step1_v2(G["q"], G["k"], G["index1"], G["offsets"], G["n_max"])
    a_g = be.attention_step1_v2(D(G["q"]), D(G["k"]), D(G["index1"]), D(G["offsets"]), G["n_max"]).cpu()
    b_o = ob.dot_prod_with_idx_v3(G["q"], G["offsets"], G["n_max"], G["k"], G["index1"], G["tq"], G["tk"], G["rel_idx"])
    b_g = be.dot_prod_with_idx_v3(D(G["q"]), D(G["offsets"]), G["n_max"], D(G["k"]), D(G["index1"]), D(G["tq"]), D(G["tk"]), D(G["rel_idx"])).cpu()
    s_o = ob.segment_softmax(a_o + b_o, G["offsets"])
    s_g = be.segment_softmax(D(a_o + b_o), D(G["offsets"])).cpu()
    x_o = ob.attention_step2_with_rel_pos_value_v2(s_o, G["v"], G["offsets"], G["n_max"], G["index1"], G["tv"], G["rel_idx"])
    x_g = be.attention_step2_with_rel_pos_value_v2(D(s_o), D(G["v"]), D(G["offsets"]), G["n_max"], D(G["index1"]), D(G["tv"]), D(G["rel_idx"])).cpu()
    print((h, d, L, deg, n), "M", G["m"], "step1", helpers.max_rel(a_g.numpy(), a_o.numpy()), "dot", helpers.max_rel(b_g.numpy(), b_o.numpy()),
          "softmax", helpers.max_rel(s_g.numpy(), s_o.numpy()), "step2", helpers.max_rel(x_g.numpy(), x_o.numpy()))
