"""CPU check (float64, torch autograd's double backward as the judge) of the algebra behind the Wasserstein gradient penalty
through the attention context network (engine.AttentionContext.tangent / gp_*; csrc/attn.hip: ln_tangent, ln_gp, attn_tangent,
attn_gp kernels).  D(theta) = <v, d S / d z> is the directional derivative of the summed scores S along v; its gradient is the
reverse sweep of the joint (primal, tangent) program: delta = adjoint of S (also the adjoint of the tangent variables),
nu = what the primal variables gain through the tangent program's coefficients (sources at LayerNorm, softmax, Q.K, P.V).

Run:  python tools/gp_attention_algebra.py        (prints relative errors, all ~1e-15)"""
import math
import torch

torch.manual_seed(0)
torch.set_default_dtype(torch.float64)
B, S, C, NH, FF, H, NL = 2, 5, 8, 2, 12, 6, 2
d = C // NH
EPS = 1e-5
P_DROP = 0.3


def mk(*shape, s=0.5):
    return (torch.randn(*shape) * s).requires_grad_(True)


layers = [dict(w_in=mk(3 * C, C), b_in=mk(3 * C), w_o=mk(C, C), b_o=mk(C), g1=mk(C, s=1.0), be1=mk(C), w1=mk(FF, C), b1=mk(FF),
               w2=mk(C, FF), b2=mk(C), g2=mk(C, s=1.0), be2=mk(C)) for _ in range(NL)]
gN, bN = mk(C, s=1.0), mk(C)
w_end, b_end = mk(H, C), mk(H)
pe = torch.randn(S, C)
z = torch.randn(B, S, C, requires_grad=True)
zt = torch.randn(B, S, C)
wc = torch.randn(B, H)
scale_z = math.sqrt(C)
masks = [dict(att=(torch.rand(B, NH, S, S) > P_DROP).double() / (1 - P_DROP), d1=(torch.rand(B, S, C) > P_DROP).double() / (1 - P_DROP),
              ff=(torch.rand(B, S, FF) > P_DROP).double() / (1 - P_DROP), d2=(torch.rand(B, S, C) > P_DROP).double() / (1 - P_DROP))
         for _ in range(NL)]
causal = torch.tril(torch.ones(S, S)).bool()


def ln(x, g, b):
    mu = x.mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((x - mu) ** 2).mean(-1, keepdim=True) + EPS)
    return (x - mu) * rstd * g + b


def heads(x):   # (B,S,C) -> (B,NH,S,d)
    return x.view(B, S, NH, d).transpose(1, 2)


def unheads(x):
    return x.transpose(1, 2).reshape(B, S, C)


def model(z):
    x = z * scale_z + pe
    for L, m in zip(layers, masks):
        qkv = x @ L["w_in"].T + L["b_in"]
        q, k, v = (heads(t) for t in qkv.split(C, -1))
        s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
        s = s.masked_fill(~causal, float("-inf"))
        p = torch.softmax(s, -1)
        o = unheads((p * m["att"]) @ v)
        y = o @ L["w_o"].T + L["b_o"]
        x1 = ln(x + y * m["d1"], L["g1"], L["be1"])
        f1 = torch.relu(x1 @ L["w1"].T + L["b1"]) * m["ff"]
        y2 = f1 @ L["w2"].T + L["b2"]
        x = ln(x1 + y2 * m["d2"], L["g2"], L["be2"])
    xn = ln(x, gN, bN)
    return xn.mean(1) @ w_end.T + b_end


params = [t for L in layers for t in L.values()] + [gN, bN, w_end, b_end]
c = model(z)
S_ = (wc * c).sum()
g, = torch.autograd.grad(S_, z, create_graph=True)
D = (g * zt).sum()
ref = torch.autograd.grad(D, params + [z], allow_unused=True)
ref = [r if r is not None else torch.zeros_like(p_) for r, p_ in zip(ref, params + [z])]

# ===================================================================== manual: the kernels' decomposition
with torch.no_grad():
    def ln_stats(r):
        mu = r.mean(-1, keepdim=True)
        rstd = 1.0 / torch.sqrt(((r - mu) ** 2).mean(-1, keepdim=True) + EPS)
        return (r - mu) * rstd, rstd

    def proj(u, xh):          # P u = u - <u> - xh <xh u>
        return u - u.mean(-1, keepdim=True) - xh * (xh * u).mean(-1, keepdim=True)

    def ln_tangent(rt, xh, rstd, w):
        return w * rstd * proj(rt, xh)

    def ln_bwd(dy, xh, rstd, w):       # -> d r, d w, d b
        return rstd * proj(dy * w, xh), (dy * xh).sum((0, 1)), dy.sum((0, 1))

    def ln_gp(dy, rt, xh, rstd, w):    # -> source on r, penalty part of d w
        p = dy * w
        pr, pp = proj(rt, xh), proj(p, xh)
        a, b = (p * xh).mean(-1, keepdim=True), (xh * rt).mean(-1, keepdim=True)
        src = -(rstd ** 2) * (xh * (p * pr).mean(-1, keepdim=True) + b * pp + a * pr)
        return src, (dy * rstd * pr).sum((0, 1))

    def attn_bwd(dO, q, k, v, p, m):   # ordinary: -> dq, dk, dv (head layout)
        a = m * (dO @ v.transpose(-1, -2))
        ds = p * (a - (p * a).sum(-1, keepdim=True)) / math.sqrt(d)
        return ds @ k, ds.transpose(-1, -2) @ q, (p * m).transpose(-1, -2) @ dO

    def attn_tangent(q, k, v, qt, kt, vt, p, m):
        u = (qt @ k.transpose(-1, -2) + q @ kt.transpose(-1, -2)) / math.sqrt(d)
        u = u.masked_fill(~causal, 0.0)
        pt = p * (u - (p * u).sum(-1, keepdim=True))
        return (pt * m) @ v + (p * m) @ vt

    def attn_gp(dO, q, k, v, qt, kt, vt, p, m):    # sources on q, k, v
        sc = 1.0 / math.sqrt(d)
        u = ((qt @ k.transpose(-1, -2) + q @ kt.transpose(-1, -2)) * sc).masked_fill(~causal, 0.0)
        mrow = (p * u).sum(-1, keepdim=True)
        pt = p * (u - mrow)
        a = m * (dO @ v.transpose(-1, -2))
        cc = (p * a).sum(-1, keepdim=True)
        dS = p * (a - cc)
        e = m * (dO @ vt.transpose(-1, -2))
        w = e + (a - cc) * (u - mrow)
        sig = p * (w - (p * w).sum(-1, keepdim=True))
        return sc * (sig @ k + dS @ kt), sc * (sig.transpose(-1, -2) @ q + dS.transpose(-1, -2) @ qt), (pt * m).transpose(-1, -2) @ dO

    # ---- primal forward, keeping what the engine keeps
    x = z * scale_z + pe
    keep = []
    for L, m in zip(layers, masks):
        K_ = dict(x=x)
        qkv = x @ L["w_in"].T + L["b_in"]
        q, k, v = (heads(t) for t in qkv.split(C, -1))
        s = ((q @ k.transpose(-1, -2)) / math.sqrt(d)).masked_fill(~causal, float("-inf"))
        p = torch.softmax(s, -1)
        att = unheads((p * m["att"]) @ v)
        y = att @ L["w_o"].T + L["b_o"]
        r1 = x + y * m["d1"]
        xh1, rs1 = ln_stats(r1)
        x1 = xh1 * L["g1"] + L["be1"]
        f1 = torch.relu(x1 @ L["w1"].T + L["b1"]) * m["ff"]          # stored dropped, as in the engine
        y2 = f1 @ L["w2"].T + L["b2"]
        r2 = x1 + y2 * m["d2"]
        xh2, rs2 = ln_stats(r2)
        x = xh2 * L["g2"] + L["be2"]
        K_.update(q=q, k=k, v=v, p=p, att=att, xh1=xh1, rs1=rs1, x1=x1, f1=f1, xh2=xh2, rs2=rs2)
        keep.append(K_)
    xhN, rsN = ln_stats(x)
    xn = xhN * gN + bN
    mean = xn.mean(1)
    # ---- tangent pass
    xt = zt * scale_z
    for L, m, K_ in zip(layers, masks, keep):
        K_["xt"] = xt
        qkvt = xt @ L["w_in"].T
        qt, kt, vt = (heads(t) for t in qkvt.split(C, -1))
        attt = unheads(attn_tangent(K_["q"], K_["k"], K_["v"], qt, kt, vt, K_["p"], m["att"]))
        r1t = xt + (attt @ L["w_o"].T) * m["d1"]
        x1t = ln_tangent(r1t, K_["xh1"], K_["rs1"], L["g1"])
        f1t = (x1t @ L["w1"].T) * (K_["f1"] != 0) * m["ff"]           # mask of the stored activation x dropout factor
        r2t = x1t + (f1t @ L["w2"].T) * m["d2"]
        xt = ln_tangent(r2t, K_["xh2"], K_["rs2"], L["g2"])
        K_.update(qt=qt, kt=kt, vt=vt, attt=attt, r1t=r1t, x1t=x1t, f1t=f1t, r2t=r2t)
    xNt = xt
    meant = ln_tangent(xNt, xhN, rsN, gN).mean(1)
    # ---- joint reverse sweep: delta (seed wc) and nu (seed 0, sources on the way)
    G = {}

    def acc(name, val):
        G[name] = G.get(name, 0) + val

    dl_c, nu_c = wc, torch.zeros_like(wc)
    acc("w_end", dl_c.T @ meant + nu_c.T @ mean)
    acc("b_end", nu_c.sum(0))
    dl = (dl_c @ w_end)[:, None, :].expand(B, S, C) / S
    nu = (nu_c @ w_end)[:, None, :].expand(B, S, C) / S
    src, gw = ln_gp(dl, xNt, xhN, rsN, gN)
    nu_r, nw, nb = ln_bwd(nu, xhN, rsN, gN)
    acc("gN", gw + nw); acc("bN", nb)
    dl, _, _ = ln_bwd(dl, xhN, rsN, gN)
    nu = nu_r + src
    for li in range(NL - 1, -1, -1):
        L, m, K_ = layers[li], masks[li], keep[li]
        # norm2
        src, gw = ln_gp(dl, K_["r2t"], K_["xh2"], K_["rs2"], L["g2"])
        nu_r, nw, nb = ln_bwd(nu, K_["xh2"], K_["rs2"], L["g2"])
        dl_r, _, _ = ln_bwd(dl, K_["xh2"], K_["rs2"], L["g2"])
        nu_r = nu_r + src
        acc((li, "g2"), gw + nw); acc((li, "be2"), nb)
        dl_y2, nu_y2 = dl_r * m["d2"], nu_r * m["d2"]
        acc((li, "w2"), dl_y2.reshape(-1, C).T @ K_["f1t"].reshape(-1, FF) + nu_y2.reshape(-1, C).T @ K_["f1"].reshape(-1, FF))
        acc((li, "b2"), nu_y2.sum((0, 1)))
        fm = (K_["f1"] != 0) * m["ff"]
        dl_f, nu_f = (dl_y2 @ L["w2"]) * fm, (nu_y2 @ L["w2"]) * fm
        acc((li, "w1"), dl_f.reshape(-1, FF).T @ K_["x1t"].reshape(-1, C) + nu_f.reshape(-1, FF).T @ K_["x1"].reshape(-1, C))
        acc((li, "b1"), nu_f.sum((0, 1)))
        dl_x1, nu_x1 = dl_r + dl_f @ L["w1"], nu_r + nu_f @ L["w1"]
        # norm1
        src, gw = ln_gp(dl_x1, K_["r1t"], K_["xh1"], K_["rs1"], L["g1"])
        nu_r, nw, nb = ln_bwd(nu_x1, K_["xh1"], K_["rs1"], L["g1"])
        dl_r, _, _ = ln_bwd(dl_x1, K_["xh1"], K_["rs1"], L["g1"])
        nu_r = nu_r + src
        acc((li, "g1"), gw + nw); acc((li, "be1"), nb)
        dl_y, nu_y = dl_r * m["d1"], nu_r * m["d1"]
        acc((li, "w_o"), dl_y.reshape(-1, C).T @ K_["attt"].reshape(-1, C) + nu_y.reshape(-1, C).T @ K_["att"].reshape(-1, C))
        acc((li, "b_o"), nu_y.sum((0, 1)))
        dl_O, nu_O = heads(dl_y @ L["w_o"]), heads(nu_y @ L["w_o"])
        dq, dk, dv = attn_bwd(dl_O, K_["q"], K_["k"], K_["v"], K_["p"], m["att"])
        nq, nk, nv = attn_bwd(nu_O, K_["q"], K_["k"], K_["v"], K_["p"], m["att"])
        sq, sk, sv = attn_gp(dl_O, K_["q"], K_["k"], K_["v"], K_["qt"], K_["kt"], K_["vt"], K_["p"], m["att"])
        dl_qkv = torch.cat([unheads(t) for t in (dq, dk, dv)], -1)
        nu_qkv = torch.cat([unheads(t) for t in (nq + sq, nk + sk, nv + sv)], -1)
        acc((li, "w_in"), dl_qkv.reshape(-1, 3 * C).T @ K_["xt"].reshape(-1, C) + nu_qkv.reshape(-1, 3 * C).T @ K_["x"].reshape(-1, C))
        acc((li, "b_in"), nu_qkv.sum((0, 1)))
        dl, nu = dl_r + dl_qkv @ L["w_in"], nu_r + nu_qkv @ L["w_in"]
    gz = nu * scale_z

mine = [G[(li, n)] for li in range(NL) for n in layers[li].keys()] + [G["gN"], G["bN"], G["w_end"], G["b_end"], gz]
names = [f"{li}.{n}" for li in range(NL) for n in layers[li].keys()] + ["gN", "bN", "w_end", "b_end", "z"]
worst = 0.0
for n, a, b in zip(names, ref, mine):
    err = ((a - b).norm() / (a.norm() + 1e-30)).item() if a.norm() > 1e-12 else b.norm().item()
    worst = max(worst, err)
    print(f"{n:10s} {err:.2e}")
print("worst", worst)
assert worst < 1e-10
