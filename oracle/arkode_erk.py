"""ARKode's default explicit integrator restated around the oracle's f().  TEST INFRASTRUCTURE ONLY (see crd_oracle.py).

What the reference runs (/root/reference/src/FHNmodel_torus.cpp:356-372,420-435): ARKodeCreate; ARKodeInit(mem, f, NULL, T0, y)
-- fi == NULL selects a purely explicit method, default order 4 --; ARKodeSStolerances(1e-5, 1e-10); ARKodeSetMaxNumSteps(200000);
then one ARKode(mem, tout, y, &t, ARK_NORMAL) call per output time.  SUNDIALS is a third-party dependency that is NOT in the
reference tree and NOT in this image; the version is unpinned (CMake/FindSUNDIALS.cmake:4 just searches $HOME/sundials), but the
ARKodeCreate / ARKodeInit(fe, fi, ...) API and the included arkode/arkode_pcg.h (:51) only exist together in SUNDIALS 2.6.x - 2.7.0
(ARKode 1.0.x - 1.1.0).  This module restates the PUBLISHED algorithm of that release line -- the ARKode documentation's
"Mathematical Considerations" chapter (time step adaptivity: the PID controller and its safeguards; initial step estimation;
Hermite dense output) and its Butcher-table appendix (ARK_ZONNEVELD_5_3_4, "Zonneveld-5-3-4", the default 4th-order ERK table)
-- recalled, because there is no network here and no SUNDIALS source to read: PARITY UNPINNED.  What pins the table: the order
conditions (tests/test_oracle.py checks them to third order for the embedding, fourth for the method, and the row sums); what
pins the controller: nothing but the documentation's formulas as restated below, each with the constant's documented name.

The method.  s = 5 stages, method order q = 4, embedding order p = 3:

      0   |
     1/2  | 1/2
     1/2  |  0    1/2
      1   |  0     0     1
     3/4  | 5/32  7/32  13/32  -1/32
     -----+---------------------------------
      b   | 1/6   1/3   1/3    1/6     0          (the classical RK4 weights: the propagated solution IS classical RK4)
      b^  | -1/2  7/3   7/3   13/6   -16/3        (third-order embedding)

so y_{n+1} = y_n + h sum b_i k_i, and the local error estimate is y_{n+1} - y^_{n+1} = h sum (b_i - b^_i) k_i
= h (2/3 k1 - 2 k2 - 2 k3 - 2 k4 + 16/3 k5), measured in the WRMS norm with weights 1 / (rtol |y_n| + atol) recomputed from the
current solution before every step (ARKodeSStolerances' built-in efun).

Step size control (defaults of that release line):
  * error test: dsm = ||error||_WRMS <= 1 accepts.
  * PID controller (ARK_ADAPT_PID, method 0): h_acc = h e1^(-k1/p) e2^(k2/p) e3^(-k3/p), k1 = 0.58, k2 = 0.21, k3 = 0.1,
    p = the EMBEDDING order (pq = 0), e1 = max(bias dsm, 1e-10) of this step, e2 / e3 of the two accepted steps before (history
    initialised to 1), bias = 1.5.
  * then: h_acc *= safety (0.96); growth limited to etamax (10000 on the very first step, "etamx1"; 20 afterwards, "growth";
    1 for the rest of a step that has failed its error test); reduction limited to ETAMIN = 0.1; explicit-stability bound (none
    by default: ARKodeSetStabilityFn was not called); and no change at all when 1 <= h_acc/h <= 1.5 ("lbound" / "ubound", a
    dead band that keeps h constant for modest suggested growth).
  * after a failed error test: the same formula with the failed step's error in front of the history (history itself not
    advanced), eta additionally <= etamxf = 0.3 from the second failure of a step on (small_nef = 2), at most maxnef = 7
    failures per step; the accepted step that follows a failure keeps its size (eta = 1: "defer step size changes").
  * ARK_NORMAL: steps are never shortened for an output time; the call returns the degree-3 Hermite interpolant
    (ARKodeGetDky, dense order min(q - 1, 3) = 3) once t_n has reached or passed tout, and the next call carries on from t_n.
  * initial step (arkHin, inherited from CVODE's cvHin): bounds hlb = 100 uround max(|t0|, |tout|), hub = min(0.1 |tout - t0|,
    1 / max_i(|f_i| / (0.1 |y_i| + rtol |y_i| + atol))); start at the geometric mean, up to 4 passes of
    ydd = (f(t0 + hg, y + hg f) - f) / hg, hnew = sqrt(2 / ||ydd||_WRMS) (or sqrt(hg hub) if ||ydd|| hub^2 <= 2), stop when
    hnew / hg is within (1/2, 2); h0 = 0.5 hnew clipped to [hlb, hub].

The stage that costs: f is evaluated 5 times per attempt (no stage is reused) plus once per accepted step at the new point for
the interpolant (the reference's ARKode 1.x evaluates f(t_{n+1}, y_{n+1}) for dense output; here it is only evaluated when an
output actually falls into the step -- same numbers, fewer calls).
"""
import math

import numpy as np

from . import crd_oracle as co

# ARK_ZONNEVELD_5_3_4 (ARKode documentation, appendix "Butcher tables", explicit tables; the default for order 4)
ZONNEVELD_5_3_4 = dict(
    c=[0.0, 0.5, 0.5, 1.0, 0.75],
    A=[[0.0, 0.0, 0.0, 0.0, 0.0],
       [0.5, 0.0, 0.0, 0.0, 0.0],
       [0.0, 0.5, 0.0, 0.0, 0.0],
       [0.0, 0.0, 1.0, 0.0, 0.0],
       [5.0 / 32.0, 7.0 / 32.0, 13.0 / 32.0, -1.0 / 32.0, 0.0]],
    b=[1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0, 0.0],
    b2=[-0.5, 7.0 / 3.0, 7.0 / 3.0, 13.0 / 6.0, -16.0 / 3.0],
    q=4, p=3)

# controller constants, by their ARKode names
ADAPT_K1, ADAPT_K2, ADAPT_K3 = 0.58, 0.21, 0.1   # ARK_ADAPT_PID defaults
SAFETY, BIAS, GROWTH = 0.96, 1.5, 20.0           # "safety", "bias", "growth" (etamax after the first step)
ETAMX1, ETAMXF, ETAMIN = 10000.0, 0.3, 0.1       # first-step growth bound, bound after repeated failures, largest reduction
SMALL_NEF, MAXNEF = 2, 7
HFIXED_LB, HFIXED_UB = 1.0, 1.5                  # dead band: no change of h for lb <= eta <= ub
ONEPSM, ONEMSM = 1.000001, 0.999999
TINY = 1.0e-10
UROUND = 2.220446049250313e-16
H0_LBFACTOR, H0_UBFACTOR, H0_BIAS, H0_ITERS = 100.0, 0.1, 0.5, 4


def wrms(v, w):
    """N_VWrmsNorm: sqrt(sum((v w)^2) / N)."""
    e = v * w
    return float(math.sqrt(float(np.sum(e * e)) / e.size))


def error_weights(y, rtol, atol):
    """ARKodeSStolerances' weights, recomputed from the current solution before every step."""
    return 1.0 / (rtol * np.abs(y) + atol)


def pid_eta(h, ehist, etamax, h_max=float("inf")):
    """arkAdapt with the PID method and no stability function: eta = h_new / h from the (biased) error history
    ehist = [this step, previous accepted, the one before]; h_max: ARKodeSetMaxStep (hmax_inv = 1 / h_max)."""
    p = ZONNEVELD_5_3_4["p"]
    e1, e2, e3 = (max(e, TINY) for e in ehist)
    h_acc = h * e1 ** (-ADAPT_K1 / p) * e2 ** (ADAPT_K2 / p) * e3 ** (-ADAPT_K3 / p)
    h_acc *= SAFETY
    h_acc = min(abs(h_acc), abs(etamax * h))
    h_acc = max(abs(h_acc), abs(ETAMIN * h))
    if abs(h_acc) > abs(h * HFIXED_LB * ONEMSM) and abs(h_acc) < abs(h * HFIXED_UB * ONEPSM):
        h_acc = h
    eta = h_acc / h
    if math.isfinite(h_max):
        eta /= max(1.0, abs(h) * eta / h_max)
    return eta


def erk_attempt(rhs, t, y, h, tab=ZONNEVELD_5_3_4):
    """One attempt: the five stages, y_new and the error vector h sum (b - b^) k (not yet normed)."""
    k = []
    for i in range(5):
        z = y
        if i > 0:
            inc = None
            for j in range(i):
                a = tab["A"][i][j]
                if a != 0.0:
                    inc = a * k[j] if inc is None else inc + a * k[j]
            z = y + h * inc
        k.append(rhs(t + tab["c"][i] * h, z))
    ynew = y + (h / 6.0) * (k[0] + 2.0 * k[1] + 2.0 * k[2] + k[3])
    err = h * ((2.0 / 3.0) * k[0] - 2.0 * (k[1] + k[2] + k[3]) + (16.0 / 3.0) * k[4])
    return ynew, err


def hermite_arkode(tau, h, yold, ynew, fold, fnew):
    """arkDenseEval, degree 3, derivative 0: tau = (t - t_n) / h in [-1, 0] measured from the END of the step."""
    a0 = 3.0 * tau * tau + 2.0 * tau * tau * tau
    a1 = 1.0 - a0
    a2 = h * tau * tau * (tau + 1.0)
    a3 = h * tau * (tau + 1.0) * (tau + 1.0)
    return a0 * yold + a1 * ynew + a2 * fold + a3 * fnew


def initial_step(rhs, t0, y0, f0, tout, rtol, atol):
    """arkHin."""
    tdist = abs(tout - t0)
    tround = UROUND * max(abs(t0), abs(tout))
    if tdist < 2.0 * tround:
        raise ValueError("tout too close to t0")
    hlb = H0_LBFACTOR * tround
    hub_inv = float(np.max(np.abs(f0) / (H0_UBFACTOR * np.abs(y0) + (rtol * np.abs(y0) + atol))))
    hub = H0_UBFACTOR * tdist
    if hub * hub_inv > 1.0:
        hub = 1.0 / hub_inv
    hg = math.sqrt(hlb * hub)
    if hub < hlb:
        return hg
    w = error_weights(y0, rtol, atol)
    hnew_ok, hnew = False, hg
    for count in range(1, H0_ITERS + 1):
        ydd = (rhs(t0 + hg, y0 + hg * f0) - f0) * (1.0 / hg)
        yddnrm = wrms(ydd, w)
        if hnew_ok or count == H0_ITERS:
            hnew = hg
            break
        hnew = math.sqrt(2.0 / yddnrm) if yddnrm * hub * hub > 2.0 else math.sqrt(hg * hub)
        hrat = hnew / hg
        if 0.5 < hrat < 2.0:
            hnew_ok = True
        if count > 1 and hrat > 2.0:
            hnew, hnew_ok = hg, True
        hg = hnew
    return min(max(H0_BIAS * hnew, hlb), hub)


class ArkodeErk:
    """The integrator object: ARKodeCreate + ARKodeInit(f, NULL, t0, y0) + ARKodeSStolerances(rtol, atol) [+ a cap on h, which
    the reference does not set: libcrd's default cap is its RK4 stability bound].  evolve(tout) is ARKode(mem, tout, y, &t,
    ARK_NORMAL)."""

    def __init__(self, problem, t0, y0, rtol=1e-5, atol=1e-10, h_max=float("inf"), max_steps=200000, h0=0.0, nthreads=1):
        self.p, self.rtol, self.atol, self.h_max, self.max_steps, self.nthreads = problem, rtol, atol, h_max, max_steps, nthreads
        self.tn = float(t0)
        self.y = np.array(y0, dtype=np.float64, order="C", copy=True)
        self.yold, self.told = None, None
        self.h, self.hprime, self.eta = float(h0), float(h0), 1.0
        self.etamax = ETAMX1
        self.ehist = [1.0, 1.0, 1.0]
        self.nst = self.nst_attempts = self.netf = self.nfe = 0
        self.steps = []  # accepted step sizes
        self._f_cache = {}

    def rhs(self, t, y):
        self.nfe += 1
        return co.rhs(self.p, t, y, nthreads=self.nthreads)

    def _cap(self, h):
        return min(h, self.h_max)

    def evolve(self, tout):
        """Returns (y(tout), stats of this call)."""
        tout = float(tout)
        st = dict(accepted=0, rejected=0)
        if self.nst == 0:
            if self.h == 0.0:
                f0 = self.rhs(self.tn, self.y)
                self.h = initial_step(self.rhs, self.tn, self.y, f0, tout, self.rtol, self.atol)
            self.h = self._cap(self.h)
            self.hprime = self.h
        elif (self.tn - tout) * self.h >= 0.0:  # tout lies in the step already taken: interpolate again
            st.update(self._stats())
            return self._dense(tout), st
        nstloc = 0
        while True:
            if nstloc >= self.max_steps:
                raise RuntimeError("ARK_TOO_MUCH_WORK: mxstep steps taken before reaching tout")
            if self.nst > 0 and self.hprime != self.h:
                self.h = self.h * self.eta
            w = error_weights(self.y, self.rtol, self.atol)
            nef = 0
            while True:  # attempts at this step
                self.nst_attempts += 1
                ynew, err = erk_attempt(self.rhs, self.tn, self.y, self.h)
                dsm = wrms(err, w)
                if dsm <= 1.0:
                    break
                nef += 1
                self.netf += 1
                st["rejected"] += 1
                if nef == MAXNEF:
                    raise RuntimeError("ARK_ERR_FAILURE: error test failed repeatedly")
                self.etamax = 1.0
                eta = pid_eta(self.h, [dsm * BIAS if math.isfinite(dsm) else 1e300, self.ehist[0], self.ehist[1]], self.etamax, self.h_max)
                if nef >= SMALL_NEF:
                    eta = min(eta, ETAMXF)
                self.h *= eta
            # arkPrepareNextStep: history first, then either "defer" (after a failure in this step) or the controller
            self.ehist = [dsm * BIAS, self.ehist[0], self.ehist[1]]
            if self.etamax == 1.0:
                self.etamax = GROWTH
                self.hprime, self.eta = self.h, 1.0
            else:
                self.eta = pid_eta(self.h, self.ehist, self.etamax, self.h_max)
                self.hprime = self.h * self.eta
            # arkCompleteStep
            self.told, self.yold = self.tn, self.y
            self.tn, self.y = self.tn + self.h, ynew
            self.nst += 1
            nstloc += 1
            st["accepted"] += 1
            self.steps.append(self.h)
            self.etamax = GROWTH
            self._f_cache = {}
            if (self.tn - tout) * self.h >= 0.0:
                st.update(self._stats())
                return self._dense(tout), st

    def _stats(self):
        return dict(t_internal=self.tn, h_last=self.h, h_next=self.h * self.eta if self.hprime != self.h else self.h, nst=self.nst, netf=self.netf)

    def _dense(self, tout):
        if tout == self.tn:
            return self.y.copy()
        h = self.tn - self.told
        if "fold" not in self._f_cache:
            self._f_cache["fold"] = self.rhs(self.told, self.yold)
            self._f_cache["fnew"] = self.rhs(self.tn, self.y)
        return hermite_arkode((tout - self.tn) / h, h, self.yold, self.y, self._f_cache["fold"], self._f_cache["fnew"])
