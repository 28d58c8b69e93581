"""Debug aid: stage-by-stage relative difference HIP vs oracle (bf16 rounding mode) on a golden fixture."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import vf_oracle as O
from tests.conftest import load_fixture
from tests.helpers import build_model
from variantformer_amd import ops
from variantformer_amd.seq2gene.modules.layers import packed_linear

def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return "max %.2e mean %.2e" % (float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)), float(np.abs(a - b).mean() / (np.abs(b).mean() + 1e-30)))

name = sys.argv[1] if len(sys.argv) > 1 else "small_sin"
meta, arrays, sd, batch = load_fixture(name)
hp = O.Seq2RegHP.from_hparams(meta["seq2reg"]); ghp = O.Seq2GeneHP.from_kwargs(meta["seq2gene"])
col = {}
with torch.no_grad():
    pred_o, emb_o = O.forward(batch, sd, hp, hp, ghp, rounding="bf16", share_cre_stream=True, collect=col)
col32 = {}
with torch.no_grad():
    pred_32, emb_32 = O.forward(batch, sd, hp, hp, ghp, rounding=None, share_cre_stream=True, collect=col32)
model = build_model(meta["seq2reg"], meta["seq2gene"], sd).cuda()
pb = model.prepare_batch(batch, dedupe_windows=False).wait()    # rows align with the batch's windows; upload finished for every reader below
with torch.no_grad():
    cre_tok = model.cre_tokenizer.embed_packed(pb.cre_ids, pb.cre_pad, pb.cre_tokens, torch.float32)
    gene_tok = model.gene_tokenizer.embed_packed(pb.gene_ids, pb.gene_pad, pb.gene_tokens, torch.float32)
    print("cre_tok  hip-vs-oracle16", rel(cre_tok.cpu(), torch.cat(col["cre_tok"])), " oracle16-vs-32", rel(torch.cat(col["cre_tok"]), torch.cat(col32["cre_tok"])))
    print("gene_tok hip-vs-oracle16", rel(gene_tok.cpu(), torch.cat(col["gene_tok"])), " oracle16-vs-32", rel(torch.cat(col["gene_tok"]), torch.cat(col32["gene_tok"])))
    # single seq2reg layer check
    cu = ops.mask_to_cu_seqlens(pb.cre_pad)
    x0 = ops.embed_pack(pb.cre_ids, pb.cre_pad, cu, model.cre_tokenizer.token_embedding.weight, model.cre_tokenizer._pos_table(pb.cre_ids.device), pb.cre_tokens)
    ids = torch.cat([v[:,0,:] for v in batch["cre_sequences"]]); pad = torch.cat([v[:,0,:] for v in batch["cre_attention_masks"]])
    xo = sd["cre_tokenizer.token_embedding.weight"][ids]
    if hp.positional_encoding == "sinusoidal": xo = xo + O.positional_encoding_1d(hp.embedding_dim, hp.token_length)
    xop, idx, cuo, _, _ = O.unpad_input(xo, ~pad)
    print("embed", rel(x0.cpu(), xop))
    rnd = O.Rounding("bf16")
    slopes = None if hp.positional_encoding == "sinusoidal" else torch.tensor(O.alibi_slopes(hp.num_heads))
    pfx = "cre_tokenizer.transformer_encoder.0."
    lay = model.cre_tokenizer.transformer_encoder[0]
    h_h = ops.layernorm(x0, lay.norm1.weight, lay.norm1.bias)
    h_o = rnd.r(O.layer_norm(xop, sd[pfx+"norm1.weight"], sd[pfx+"norm1.bias"]))
    print(" ln1", rel(h_h.float().cpu(), h_o), "mismatch frac", float((h_h.float().cpu()!=h_o).float().mean()))
    w, b = packed_linear(lay.MHA.Wqkv)
    qkv_h = ops.gemm(h_h, w, b, ops.EPI_BF16)
    qkv_o = rnd.r(O.linear(h_o, sd[pfx+"MHA.Wqkv.weight"], sd[pfx+"MHA.Wqkv.bias"], rnd))
    print(" qkv", rel(qkv_h.float().cpu(), qkv_o), "mismatch frac", float((qkv_h.float().cpu()!=qkv_o).float().mean()))
    a_h = lay.MHA.attend(h_h, None, cu, 200, None, None)
    D = hp.embedding_dim; H = hp.num_heads
    a_o = torch.zeros_like(qkv_o[:, :D])
    q3 = qkv_o.view(-1, 3, H, D//H)
    for bb in range(len(cuo)-1):
        s, e = int(cuo[bb]), int(cuo[bb+1])
        a_o[s:e] = O.attention(q3[s:e,0], q3[s:e,1], q3[s:e,2], slopes, rnd).reshape(e-s, D)
    a_o = rnd.r(a_o)
    print(" attn", rel(a_h.float().cpu(), a_o), "mismatch frac", float((a_h.float().cpu()!=a_o).float().mean()))
    w, b = packed_linear(lay.MHA.out_proj)
    y1_h = ops.gemm(a_h, w, b, ops.EPI_RES_F32, residual=x0)
    y1_o = O.linear(a_o, sd[pfx+"MHA.out_proj.weight"], sd[pfx+"MHA.out_proj.bias"], rnd) + xop
    print(" x1 (out_proj+res)", rel(y1_h.cpu(), y1_o))
    y1_same = ops.gemm(a_o.cuda().bfloat16(), w, b, ops.EPI_RES_F32, residual=xop.cuda())
    print(" x1 with identical inputs", rel(y1_same.cpu(), y1_o))
    h2_h = ops.layernorm(y1_h, lay.norm2.weight, lay.norm2.bias)
    h2_o = rnd.r(O.layer_norm(y1_o, sd[pfx+"norm2.weight"], sd[pfx+"norm2.bias"]))
    print(" ln2", rel(h2_h.float().cpu(), h2_o), "mismatch frac", float((h2_h.float().cpu()!=h2_o).float().mean()))
    w1, b1 = packed_linear(lay.linear_geglu_1, geglu=True)
    hg_h = ops.gemm(h2_h, w1, b1, ops.EPI_GEGLU_BF16)
    hh = O.linear(h2_o, sd[pfx+"linear_geglu_1.weight"], sd[pfx+"linear_geglu_1.bias"], rnd)
    aa, gg = hh.chunk(2, dim=-1)
    hg_o = rnd.r(aa * torch.nn.functional.gelu(gg))
    print(" geglu hidden", rel(hg_h.float().cpu(), hg_o), "mismatch frac", float((hg_h.float().cpu()!=hg_o).float().mean()))
    hg_same = ops.gemm(h2_o.cuda().bfloat16(), w1, b1, ops.EPI_GEGLU_BF16)
    print(" geglu hidden identical inputs", rel(hg_same.float().cpu(), hg_o), "mismatch frac", float((hg_same.float().cpu()!=hg_o).float().mean()))
    w2, b2 = packed_linear(lay.linear_geglu_2)
    o_same = ops.gemm(hg_o.cuda().bfloat16(), w2, b2, ops.EPI_RES_F32, residual=xop.cuda())
    o_o = O.linear(hg_o, sd[pfx+"linear_geglu_2.weight"], sd[pfx+"linear_geglu_2.bias"], rnd) + xop
    print(" ffn out identical inputs", rel(o_same.cpu(), o_o))
    x1_h = lay.forward_packed(x0, cu, 200)
    x1_o = O.seq2reg_layer(xop, cuo, sd, pfx, hp, slopes, rnd)
    x1_32 = O.seq2reg_layer(xop, cuo, sd, pfx, hp, slopes, O.Rounding(None))
    print(" layer0 out hip-vs-o16", rel(x1_h.cpu(), x1_o), " o16-vs-o32", rel(x1_o, x1_32))
    pred, emb, gene_out, cre_out = model.forward_prepared(pb, return_cre=True)[:4]
print("first_gene_layer n/a; final gene_out hip-vs-o16", rel(gene_out.cpu(), torch.cat([m.reshape(-1, m.shape[-1]) for m in col["modulator_gene_out"]])),
      " o16-vs-o32", rel(torch.cat([m.reshape(-1, m.shape[-1]) for m in col["modulator_gene_out"]]), torch.cat([m.reshape(-1, m.shape[-1]) for m in col32["modulator_gene_out"]])))
print("emb  hip-vs-o16", rel(emb.cpu(), emb_o), " o16-vs-o32", rel(emb_o, emb_32))
print("pred hip-vs-o16", rel(pred.cpu(), pred_o), " o16-vs-o32", rel(pred_o, pred_32), " hip-vs-o32", rel(pred.cpu(), pred_32))
