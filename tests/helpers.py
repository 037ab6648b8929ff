"""Program builders shared by the CPU and GPU tests."""
import numpy as np

from genjax_amd import _abi as A
from genjax_amd.program import PackedProgram, Param, SiteList
from genjax_amd import workloads as W

KIND = dict(normal=A.NORMAL, flip=A.FLIP, bernoulli_logits=A.BERNOULLI_LOGITS, beta=A.BETA, uniform=A.UNIFORM,
            exponential=A.EXPONENTIAL, half_normal=A.HALF_NORMAL, laplace=A.LAPLACE, log_normal=A.LOG_NORMAL,
            cauchy=A.CAUCHY, gamma=A.GAMMA, mv_normal_diag=A.MVNORMAL_DIAG, student_t=A.STUDENT_T,
            truncated_normal=A.TRUNCATED_NORMAL, poisson=A.POISSON, geometric=A.GEOMETRIC, dirichlet=A.DIRICHLET, gumbel=A.GUMBEL,
            half_cauchy=A.HALF_CAUCHY, inverse_gamma=A.INVERSE_GAMMA, weibull=A.WEIBULL, logit_normal=A.LOGIT_NORMAL, chi2=A.CHI2,
            chi=A.CHI, exp_gamma=A.EXP_GAMMA, exp_inverse_gamma=A.EXP_INVERSE_GAMMA, half_student_t=A.HALF_STUDENT_T, kumaraswamy=A.KUMARASWAMY,
            moyal=A.MOYAL, truncated_cauchy=A.TRUNCATED_CAUCHY, double_sided_maxwell=A.DOUBLESIDED_MAXWELL, inverse_gaussian=A.INVERSE_GAUSSIAN,
            negative_binomial=A.NEGATIVE_BINOMIAL, von_mises=A.VON_MISES)
NPAR = dict(normal=2, flip=1, bernoulli_logits=1, beta=2, uniform=2, exponential=1, half_normal=1, laplace=2,
            log_normal=2, cauchy=2, gamma=2, mv_normal_diag=2, student_t=3, truncated_normal=4, poisson=1, geometric=1,
            dirichlet=1, gumbel=2, half_cauchy=2, inverse_gamma=2, weibull=2, logit_normal=2, chi2=1,
            chi=1, exp_gamma=2, exp_inverse_gamma=2, half_student_t=3, kumaraswamy=2, moyal=2, truncated_cauchy=4, double_sided_maxwell=2,
            inverse_gaussian=2, negative_binomial=2, von_mises=2)


def one_site(kind: str, a, b=None, obs=None, rng=A.RNG_FLAT, c=None, d=None):
    sl = SiteList()
    params = [a, b, c, d][: NPAR[kind]]
    sl.add("v", KIND[kind], params)
    if obs is None:
        return PackedProgram(sl, rng_mode=rng)
    return PackedProgram(sl, {"v": A.MODE_OBS_TAB}, {"v": obs}, rng_mode=rng)


def gmm(D=16, C=8, rng=A.RNG_FLAT, seed=0):
    return W.gmm_program(D, C, rng, seed)


def flip_flip(trivial: bool, rng=A.RNG_FLAT):
    """tests/inference/test_smc.py:32-87 of the reference."""
    sl = SiteList()
    sl.add("x", A.FLIP, [0.5])
    sl.add("y", A.FLIP, [0.7] if trivial else [Param.gather(np.array([0.3, 0.9], np.float32), "x")])
    return PackedProgram(sl, {"y": A.MODE_OBS_TAB}, {"y": 1.0}, rng_mode=rng)


def beta_bernoulli(obs: bool, rng=A.RNG_FLAT):
    """README.md:89-102 of the reference."""
    sl = SiteList()
    sl.add("p", A.BETA, [2.0, 2.0])
    sl.add("v", A.FLIP, [Param.value("p")])
    return PackedProgram(sl, {"v": A.MODE_OBS_TAB}, {"v": float(obs)}, rng_mode=rng)


def zoo(rng=A.RNG_FLAT, observed=()):
    """One site of every distribution kind, chained through every parameter-expression form."""
    sl = SiteList()
    sl.add("n0", A.NORMAL, [0.5, 2.0])
    sl.add("f0", A.FLIP, [0.3])
    sl.add("c0", A.CATEGORICAL_PROBS, [np.array([0.2, 0.5, 0.3], np.float32)])
    sl.add("n1", A.NORMAL, [Param.gather(np.array([-1.0, 0.0, 4.0], np.float32), "c0"), Param.value("n0", xf=A.XF_EXP)])
    sl.add("b0", A.BETA, [2.0, 3.5])
    sl.add("f1", A.FLIP, [Param.value("b0")])
    sl.add("u0", A.UNIFORM, [-1.0, 3.0])
    sl.add("bl", A.BERNOULLI_LOGITS, [Param.affine(np.array([[0.7]], np.float32), "u0", bias=-0.2)])
    sl.add("mv", A.MVNORMAL_DIAG, [np.array([0.0, 1.0, 2.0], np.float32), np.array([1.0, 0.5, 2.0], np.float32)], dim=3)
    sl.add("mv2", A.MVNORMAL_DIAG, [Param.affine(np.array([[1, 0, 1], [0, 2, 0]], np.float32), "mv", bias=[0.1, -0.1]),
                                    Param.const([0.7])], dim=2)
    sl.add("e0", A.EXPONENTIAL, [1.5])
    sl.add("h0", A.HALF_NORMAL, [Param.value("e0", xf=A.XF_SOFTPLUS)])
    sl.add("l0", A.LAPLACE, [Param.value("h0"), 0.8])
    sl.add("ln", A.LOG_NORMAL, [0.2, 0.4])
    sl.add("ca", A.CAUCHY, [Param.value("ln"), 2.0])
    sl.add("g0", A.GAMMA, [2.5, 1.5])
    sl.add("g1", A.GAMMA, [0.6, 2.0])
    sl.add("cl", A.CATEGORICAL_LOGITS, [Param.affine(np.array([[1.0], [0.0], [-1.0], [0.5]], np.float32), "n0")])
    sl.add("n2", A.NORMAL, [Param.value("ca", xf=A.XF_SIGMOID), 0.3])
    obs_vals = dict(n2=0.4, mv2=[0.3, 1.9], f1=1.0, l0=0.9)
    modes = {a: A.MODE_OBS_TAB for a in observed}
    return PackedProgram(sl, modes, {a: obs_vals[a] for a in observed}, rng_mode=rng)


def logreg(N=64, P=4, rng=A.RNG_FLAT, seed=0):
    return W.logreg_program(N, P, rng, seed)


def zoo2(rng=A.RNG_FLAT, observed=()):
    """The wider distribution set, one site per kind, chained through parameter forms."""
    sl = SiteList()
    sl.add("t0", A.STUDENT_T, [4.0, 0.5, 1.5])
    sl.add("tn", A.TRUNCATED_NORMAL, [Param.value("t0", xf=A.XF_SIGMOID), 0.8, -0.5, 2.0])
    sl.add("po", A.POISSON, [3.5])
    sl.add("pl", A.POISSON, [Param.value("tn", xf=A.XF_EXP)])
    sl.add("pb", A.POISSON, [40.0])                                    # transformed-rejection branch
    sl.add("ge", A.GEOMETRIC, [0.3])
    sl.add("di", A.DIRICHLET, [np.array([0.8, 2.0, 3.0], np.float32)], dim=3)
    sl.add("gu", A.GUMBEL, [Param.value("di", length=1, elem=1), 0.7])
    sl.add("hc", A.HALF_CAUCHY, [0.0, Param.value("gu", xf=A.XF_SOFTPLUS)])
    sl.add("ig", A.INVERSE_GAMMA, [3.0, 2.0])
    sl.add("we", A.WEIBULL, [1.5, Param.value("ig")])
    sl.add("lo", A.LOGIT_NORMAL, [Param.affine(np.array([[0.1]], np.float32), "po"), 0.6])
    sl.add("c2", A.CHI2, [5.0])
    sl.add("t1", A.STUDENT_T, [3.0, Param.value("lo"), Param.value("c2", xf=A.XF_SOFTPLUS)])
    sl.add("tr", A.TRUNCATED_NORMAL, [0.0, 1.0, 2.5, 6.0])            # far upper tail
    sl.add("n9", A.NORMAL, [Param.value("t1", xf=A.XF_SIGMOID), 0.3])
    obs_vals = dict(n9=0.4, di=[0.2, 0.3, 0.5], po=2.0, tn=0.7)
    modes = {a: A.MODE_OBS_TAB for a in observed}
    return PackedProgram(sl, modes, {a: obs_vals[a] for a in observed}, rng_mode=rng)


def zoo3(rng=A.RNG_FLAT, observed=()):
    """Nine more of the reference's TFP wrappers (round 6: chi, exp_gamma, exp_inverse_gamma, half_student_t, kumaraswamy, moyal,
    truncated_cauchy, double_sided_maxwell, inverse_gaussian), one site per kind plus chained copies through parameter forms."""
    sl = SiteList()
    sl.add("ch", A.CHI, [3.0])
    sl.add("eg", A.EXP_GAMMA, [2.5, 1.5])
    sl.add("ei", A.EXP_INVERSE_GAMMA, [3.0, 2.0])
    sl.add("hs", A.HALF_STUDENT_T, [5.0, 0.5, 1.5])
    sl.add("ku", A.KUMARASWAMY, [2.0, 3.0])
    sl.add("mo", A.MOYAL, [0.3, 0.8])
    sl.add("tc", A.TRUNCATED_CAUCHY, [0.2, 1.5, -2.0, 3.0])
    sl.add("dm", A.DOUBLESIDED_MAXWELL, [0.4, 0.7])
    sl.add("ig", A.INVERSE_GAUSSIAN, [1.5, 4.0])
    # chained: parameters that are values of earlier choices (under transforms that keep them in their domains)
    sl.add("ch2", A.CHI, [Param.value("ig", xf=A.XF_SOFTPLUS)])
    sl.add("eg2", A.EXP_GAMMA, [Param.value("ch"), Param.value("eg", xf=A.XF_EXP)])
    sl.add("ei2", A.EXP_INVERSE_GAMMA, [9.5, Param.value("ku", xf=A.XF_EXP)])                 # (concentration >= 8: the big-shape branch)
    sl.add("hs2", A.HALF_STUDENT_T, [Param.value("ch", xf=A.XF_SOFTPLUS), Param.value("mo"), Param.value("ig")])
    sl.add("ku2", A.KUMARASWAMY, [Param.value("ch"), Param.value("hs")])
    sl.add("mo2", A.MOYAL, [Param.value("dm"), Param.value("ku", xf=A.XF_SOFTPLUS)])
    sl.add("tc2", A.TRUNCATED_CAUCHY, [Param.value("mo", xf=A.XF_SIGMOID), Param.value("ch"), -1.0, 4.0])
    sl.add("dm2", A.DOUBLESIDED_MAXWELL, [Param.value("tc"), Param.value("ku", xf=A.XF_EXP)])
    sl.add("ig2", A.INVERSE_GAUSSIAN, [Param.value("ch"), Param.value("eg", xf=A.XF_EXP)])
    sl.add("nb", A.NEGATIVE_BINOMIAL, [4.5, 0.3])
    sl.add("vm", A.VON_MISES, [0.7, 2.5])
    sl.add("vs", A.VON_MISES, [-0.4, 0.3])                                                     # (small concentration: nearly uniform)
    sl.add("vb", A.VON_MISES, [Param.value("mo", xf=A.XF_SIGMOID), Param.value("ch", xf=A.XF_EXP)])   # (concentration up to exp(3))
    sl.add("nb2", A.NEGATIVE_BINOMIAL, [Param.value("ch"), Param.value("vm")])
    sl.add("n9", A.NORMAL, [Param.value("ig2", xf=A.XF_SIGMOID), 0.3])
    obs_vals = dict(n9=0.4, ku=0.35, mo=0.9, ig=1.1)
    modes = {a: A.MODE_OBS_TAB for a in observed}
    return PackedProgram(sl, modes, {a: obs_vals[a] for a in observed}, rng_mode=rng)


def shape_hierarchy(rng=A.RNG_FLAT):
    """a model whose SHAPE parameters are latent (gradients through them need digamma): log-normal hyper-parameters
    feeding gamma / beta concentrations, student-t / chi2 degrees of freedom and the inverse-gamma concentration"""
    from genjax_amd.program import PackedProgram, Param, SiteList
    sl = SiteList()
    sl.add("la", A.NORMAL, [0.3, 0.25])
    sl.add("lb", A.NORMAL, [0.1, 0.25])
    a, b = Param.value("la", xf=A.XF_EXP), Param.value("lb", xf=A.XF_EXP)
    sl.add("g", A.GAMMA, [a, b])
    sl.add("be", A.BETA, [a, b])
    sl.add("t", A.STUDENT_T, [Param.value("la", xf=A.XF_SOFTPLUS), 0.2, b])
    sl.add("ig", A.INVERSE_GAMMA, [a, 1.5])
    sl.add("c2", A.CHI2, [Param.value("lb", xf=A.XF_SOFTPLUS)])
    return sl


def scan_chain(T, rng=A.RNG_FLAT, carry=True, observe=False, sigma=0.1, r=0.5, seed=0, scan_id=0):
    """A Scan of T steps laid out as the host tracer does (gen.py ScanCombinator): step t holds x_t ~ N(x_{t-1}, sigma)
    (x_{-1} = 0; independent N(0, 1) draws without `carry`) and, with `observe`, y_t ~ N(x_t, r) constrained to data;
    every site of step t carries the tag GJX_SCAN_TAG(scan_id, t)."""
    rs = np.random.default_rng(seed)
    sl = SiteList()
    modes, obs = {}, {}
    ys = rs.standard_normal(T).astype(np.float32)
    for t in range(T):
        loc = Param.value(("x", t - 1)) if (carry and t > 0) else Param.const(0.0)
        sx = sl.add(("x", t), A.NORMAL, [loc, Param.const(sigma if carry else 1.0)])
        sx.scan = (scan_id << 20) | (t + 1)
        if observe:
            sy = sl.add(("y", t), A.NORMAL, [Param.value(("x", t)), Param.const(r)])
            sy.scan = (scan_id << 20) | (t + 1)
            modes[("y", t)] = A.MODE_OBS_TAB
            obs[("y", t)] = ys[t]
    return PackedProgram(sl, modes, obs, rng_mode=rng), ys


# ---------------------------------------------------------------------------------------------
# random general expressions (GJX_P_EXPR): shared by the CPU tests of the oracle and the GPU differential tests
# ---------------------------------------------------------------------------------------------
def random_expr_outs(rs, cont, dim, dom="real", depth=3):
    """`dim` output nodes (genjax_amd/expr.py) of a random elementwise expression over the elements of the earlier continuous sites
    ``cont`` = [(addr, dim), ...] and constants.  Magnitudes stay bounded (every branch passes through tanh / sigmoid / sin or a
    bounded rational form before it is combined), the final value lands in the domain the parameter needs: "real", "pos", "prob"."""
    from genjax_amd import expr as E

    def leaf():
        if rs.random() < 0.25 or not cont:
            return E.const(float(rs.standard_normal() * 0.7))
        a, d = cont[int(rs.integers(len(cont)))]
        return E.value(a, int(rs.integers(d)))

    def bounded(n):          # any node -> (-1, 1)-ish, WELL CONDITIONED: no periodic or exploding function of an unbounded argument
        r = rs.random()      # (sin(exp(a)) amplifies the last bit of the device's v_exp_f32 into an O(1) difference: not a parity question)
        if r < 0.4:
            return E.unary("tanh", n)
        if r < 0.6:
            return E.unary("sin", E.lin(0.0, [(E.unary("tanh", n), 2.0)]))
        if r < 0.8:
            return E.lin(-1.0, [(E.unary("sigmoid", n), 2.0)])
        return E.binary("div", n, E.lin(1.0, [(E.unary("square", n), 1.0)]))      # x / (1 + x^2)

    def tree(d):
        if d == 0:
            return bounded(leaf())
        r = rs.random()
        a = tree(d - 1)
        if r < 0.18:
            return E.add(a, tree(d - 1))
        if r < 0.3:
            return E.sub(a, tree(d - 1))
        if r < 0.5:
            return E.binary("mul", a, tree(d - 1))
        if r < 0.58:
            return E.binary("max" if rs.random() < 0.5 else "min", a, tree(d - 1))
        if r < 0.66:
            return E.where(E.binary("gt", a, tree(d - 1)), tree(d - 1), a)
        if r < 0.74:          # a small linear layer over several sub-expressions
            k = int(rs.integers(2, 4))
            return bounded(E.lin(float(rs.standard_normal() * 0.3), [(tree(d - 1), float(rs.standard_normal() * 0.8)) for _ in range(k)]))
        if r < 0.8:
            return E.unary("abs", a)
        if r < 0.86:
            return E.unary("log1p", E.unary("square", a))
        if r < 0.92:
            return E.unary("sqrt", E.lin(0.5, [(E.unary("square", a), 1.0)]))
        return bounded(E.unary("exp", bounded(a)))

    outs = []
    for _ in range(dim):
        n = tree(int(rs.integers(1, depth + 1)))
        if dom == "pos":
            n = E.lin(0.25, [(E.unary("softplus", n), 1.0)]) if rs.random() < 0.5 else E.unary("exp", E.lin(0.0, [(bounded(n), 0.8)]))
        elif dom == "prob":
            n = E.lin(0.02, [(E.unary("sigmoid", n), 0.96)])
        outs.append(n)
    return outs


def bnn_model(N=64, DI=16, DH=8, seed=0, first_is_one=False):
    """a small Bayesian network classifier, the canonical model of general expressions inside a vmapped kernel: latent weights outside
    the plate (one site per hidden unit's row, one for the output layer), the observations vmapped over the data —
    y_n ~ bernoulli(logits = w2 . tanh(W1 x_n)).  -> (model, X, Y, float64 log-likelihood function of (W1 [DH][DI][K], w2 [DH][K]))"""
    import genjax_amd as genjax
    rs = np.random.default_rng(seed)
    X = rs.standard_normal((N, DI)).astype(np.float32)
    if first_is_one:
        X[3, 0] = 1.0          # (a covariate that constant-folds differently in ONE instance: the plate is irregular and stays unrolled)
    Y = (rs.random(N) < 0.5).astype(np.float32)

    @genjax.gen
    def kern(x_row, W1, w2):
        h = genjax.tanh(genjax.array([genjax.dot(W1[j], x_row) for j in range(DH)]))
        return genjax.bernoulli(logits=genjax.dot(w2, h)) @ "y"

    @genjax.gen
    def bnn():
        W1 = [genjax.normal(np.zeros(DI, np.float32), 0.5) @ f"W1_{j}" for j in range(DH)]
        w2 = genjax.normal(np.zeros(DH, np.float32), 1.0) @ "w2"
        kern.vmap(in_axes=(0, None, None))(X, W1, w2) @ "obs"

    def loglik(W1, w2):
        h = np.tanh(np.einsum("jik,ni->njk", W1, X.astype(np.float64)))
        lg = np.einsum("jk,njk->nk", w2, h)
        return (Y[:, None] * -np.logaddexp(0, -lg) + (1 - Y[:, None]) * -np.logaddexp(0, lg)).sum(0)

    return bnn, X, Y, loglik
