"""Model configurations shared by the tests (small versions of BASELINE.json's configs)."""


def cfg_nif(n=8, L=2, nst=6, lst=2, r=1, si=1, so=1, pi=1, act="swish"):
    cs = {"input_dim": si, "output_dim": so, "units": n, "nlayers": L, "activation": act}
    cp = {"input_dim": pi, "latent_dim": r, "units": nst, "nlayers": lst, "activation": act}
    return "NIF", cs, cp


def cfg_ms(n=8, L=2, nst=6, lst=2, r=1, si=1, so=1, pi=1, s_res=False, p_act="sine", p_res=False,
           omega=30.0, wif=0.01, conn="full"):
    cs = {"input_dim": si, "output_dim": so, "units": n, "nlayers": L, "use_resblock": s_res,
          "connectivity": conn, "omega_0": omega, "weight_init_factor": wif}
    cp = {"input_dim": pi, "latent_dim": r, "units": nst, "nlayers": lst, "activation": p_act,
          "use_resblock": p_res, "omega_0": omega}
    return "NIFMultiScale", cs, cp


def cfg_ll(n=8, L=2, nst=6, lst=2, r=3, si=2, so=2, pi=1, s_res=False, p_act="sine", p_res=False,
           omega=30.0, wif=0.01):
    k, cs, cp = cfg_ms(n, L, nst, lst, r, si, so, pi, s_res, p_act, p_res, omega, wif, conn="last_layer")
    return "NIFMultiScaleLastLayerParameterized", cs, cp


ALL_SMALL = {
    "nif_swish": cfg_nif(),
    "nif_tanh_r2_so2": cfg_nif(r=2, so=2, si=2, act="tanh"),
    "ms_plain": cfg_ms(),
    "ms_plain_r3_si2": cfg_ms(r=3, si=2, pi=2),
    "ms_res": cfg_ms(s_res=True),
    "ms_res_pres": cfg_ms(s_res=True, p_res=True),
    "ms_mlp_pnet": cfg_ms(p_act="swish"),
    "ms_mlp_pres": cfg_ms(p_act="swish", p_res=True, so=2),
    "ll_plain": cfg_ll(),
    "ll_res": cfg_ll(s_res=True, p_res=True),
}
