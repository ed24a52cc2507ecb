/*
 * tapir_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  The product (tapir_amd/) never links, imports or falls back to it.
 *
 * What it restates (reference = /root/reference, cited file:line):
 *   orc_townsend_pi / orc_net_pi     tapir/compute.py:46-48, bin/tapir_compute.py:119 (nansum over sites)
 *   orc_quad_townsend                scipy.integrate.quad as called at tapir/compute.py:50-52:
 *                                    QUADPACK dqagse (21-point Gauss-Kronrod, epsabs=epsrel=1.49e-8,
 *                                    limit=50).  QUADPACK is third-party (netlib, public domain, shipped
 *                                    inside scipy; reference pins only scipy>=0.9.0, setup.py:27); its
 *                                    published algorithm (Piessens et al. 1983: dqagse/dqk21/dqelg/dqpsrt)
 *                                    is restated here and checked against scipy.integrate.quad in tests.
 *   orc_integral_closed              analytic antiderivative of tapir/compute.py:46-48:
 *                                    int 16 r^2 t exp(-4 r t) dt = -(4rt+1) exp(-4rt)
 *   orc_site_rates                   HyPhy stage 2, tapir/data/models_and_rates.bf:978-1070: per column,
 *                                    maximise over siteRate s>=0 the Felsenstein pruning likelihood of the
 *                                    fixed tree with every branch length multiplied by s, GTR rate matrix
 *                                    Q = R o pi (unnormalised, bf:978-1001), start s=1 (bf:1050);
 *                                    reported rate = s*kappa, subst = rate*chronoLength (bf:1056-1061),
 *                                    ll = max log L (bf:1066).  HyPhy itself is third-party, not vendored and
 *                                    not runnable here: the only pin is the known-answer file
 *                                    tapir/tests/test-hyphy/chr1_918.subsmodel.phydesign.rates (4 dp).
 *   orc_informative_counts           tapir/compute.py:96-106 (count of A/C/G/T cells per column)
 *
 * Optimiser policy (HyPhy's own derivative-free optimiser is not reproducible):
 *   maximise f(u) = log L(exp(u)) from the column's parsimony rate (fitch_start; HyPhy: s=1) with a safeguarded
 *   Newton iteration that follows the uphill direction to the nearest local maximum (on trees of 32 taxa or more the
 *   first step also uses the column's parsimony length, the slope of f as u -> -inf; maximise_column).  Flags: 0 interior optimum, 1 flat (<=1 resolved
 *   taxon: L does not depend on s, s stays 1), 2 saturated (log L flat to fp64 resolution on the way to
 *   s -> infinity, or still uphill at s = 1e4: s = 1e4 is reported), 3 optimum at s = 0 (all resolved
 *   taxa carry the same base), 4 iteration limit.
 *
 * State encoding: one byte per cell, bit mask A=1 C=2 G=4 T=8; gap/?/N = 15; IUPAC codes = unions.
 * Alignment layout: taxon-major, states[taxon*ncols + col].
 * Tree layout: nodes in post-order (children before parents, root last); parent[root] = -1;
 *   leaf_taxon[n] = row of the alignment for a leaf, -1 for an internal node; blen[n] = length of the
 *   branch above node n (ignored for the root), already divided by the correction factor.
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_U_MIN (-23.025850929940457) /* log(1e-10) */
#define ORC_U_MAX (9.210340371976184)   /* log(1e4)   */
#define ORC_STEP_MAX 2.0
#ifndef ORC_STEP_TOL
#define ORC_STEP_TOL 3e-4       /* accept when the step is this small: it is then applied with a third-order correction */
#endif
#define ORC_HALLEY_SPAN 1.0      /* the two curvatures behind the Halley step of a weakly curved point lie this close */
#define ORC_STEP_TOL_FIRST 1e-6 /* ... except at the first evaluation, where no second point exists yet */
#define ORC_MAXIT 100
#ifndef ORC_PLATEAU_STRIDE
#define ORC_PLATEAU_STRIDE 0.5
#endif
#define ORC_U_CHECK 2.995732273553991 /* log(20): optima beyond this rate are confirmed by value, see maximise_column */
#define ORC_SAT_TOL 1e-10
#ifndef ORC_HERMITE_TOL /* 0 switches the rule off (accuracy experiments against a tight reference) */
#define ORC_HERMITE_SPAN 3e-4 /* ... and only while |step| * |distance between the two points| stays below this */
#define ORC_HERMITE_TOL 2e-3 /* final step from the two-point quartic model of f' accepted below this size */
#else
#define ORC_HERMITE_SPAN 3e-4
#endif
#define ORC_HERMITE_T2D3 1.5e-8    /* step^2 * |distance of the far point|^3 below this (the quartic's error term) */
#define ORC_HERMITE_GUARD 0.005    /* the quartic's higher-order terms at the step, relative to |h| */
#define ORC_HERMITE_REGULAR 0.25   /* curvature |h| from which the two exits apply (below: iterate until the step is below 1e-6) */
#define ORC_HERMITE_NOISE 5.3e-5 /* 30 * (8 * 2.2e-16) / 1e-9, see maximise_column */
#define ORC_FIRST_STEP_MIN_TAXA 32 /* trees from this size on use the parsimony length in the first step (same constant as the engine) */
#define ORC_FLAT_EPS 1e-10 /* |dlogL/du| and |d2logL/du2| below this: surface flat to fp64 -> saturated */

/* ------------------------------------------------------------------------------------------------
 * PI(t) and its integrals
 * ---------------------------------------------------------------------------------------------- */

/* tapir/compute.py:46-48, same operation order: ((16*(r*r))*t)*exp(-((4*r)*t)) */
double orc_townsend_pi(double t, double r) { return 16.0 * (r * r) * t * exp(-(4.0 * r * t)); }

/* bin/tapir_compute.py:114,119: pi = get_townsend_pi(time_vector, rates); nansum(pi, axis=1) */
void orc_net_pi(const double *rates, int64_t n, int32_t T, double *net_out) {
    for (int32_t t = 0; t < T; ++t) {
        double acc = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double v = orc_townsend_pi((double)t, rates[i]);
            if (!isnan(v)) acc += v;
        }
        net_out[t] = acc;
    }
}

/* closed form: F(t) = -(4rt+1)exp(-4rt);  integral = F(b)-F(a) = g(4ra)-g(4rb)... written with
 * g(x) = 1-(1+x)exp(-x) (series for small x) so that tiny rates keep their digits. */
static double g_small(double x) {
    if (x < 0.1) { /* g(x) = sum_{k>=2} (-1)^k (k-1) x^k / k!  =  x^2/2 - x^3/3 + x^4/8 - ... */
        double term = x * x * 0.5, sum = 0.0; /* term = x^k / k! */
        for (int k = 2; k < 40; ++k) {
            double add = term * (double)(k - 1);
            sum += (k & 1) ? -add : add;
            if (add <= 1e-20 * fabs(sum)) break;
            term = term * x / (double)(k + 1);
        }
        return sum;
    }
    return 1.0 - (1.0 + x) * exp(-x);
}
double orc_integral_closed(double a, double b, double r) {
    if (isnan(r)) return NAN;
    return g_small(4.0 * r * b) - g_small(4.0 * r * a);
}

/* ---- QUADPACK dqk21: 21-point Gauss-Kronrod rule --------------------------------------------- */
static const double WG[5] = {0.066671344308688137593568809893332, 0.149451349150580593145776339657697,
                             0.219086362515982043995534934228163, 0.269266719309996355091226921569469,
                             0.295524224714752870173815619188769};
static const double XGK[11] = {0.995657163025808080735527280689003, 0.973906528517171720077964012084452,
                               0.930157491355708226001207180059508, 0.865063366688984510732096688423493,
                               0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
                               0.562757134668604683339000099272694, 0.433395394129247190799265943165784,
                               0.294392862701460198131126603103866, 0.148874338981631210884826001129720,
                               0.0};
static const double WGK[11] = {0.011694638867371874278064396062192, 0.032558162307964727478818972459390,
                               0.054755896574351996031381300244580, 0.075039674810919952767043140916190,
                               0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
                               0.123491976262065851077958109585166, 0.134709217311473325928054001771707,
                               0.142775938577060080797094273138717, 0.147739104901338491374841515972068,
                               0.149445554002916905664936468389821};

static void dqk21(double rate, double a, double b, double *result, double *abserr, double *resabs,
                  double *resasc) {
    const double epmach = DBL_EPSILON, uflow = DBL_MIN;
    double fv1[10], fv2[10];
    double centr = 0.5 * (a + b), hlgth = 0.5 * (b - a), dhlgth = fabs(hlgth);
    double resg = 0.0, fc = orc_townsend_pi(centr, rate), resk = WGK[10] * fc;
    *resabs = fabs(resk);
    for (int j = 0; j < 5; ++j) {
        int jtw = 2 * j + 1;
        double absc = hlgth * XGK[jtw];
        double f1 = orc_townsend_pi(centr - absc, rate), f2 = orc_townsend_pi(centr + absc, rate);
        fv1[jtw] = f1; fv2[jtw] = f2;
        double fsum = f1 + f2;
        resg += WG[j] * fsum;
        resk += WGK[jtw] * fsum;
        *resabs += WGK[jtw] * (fabs(f1) + fabs(f2));
    }
    for (int j = 0; j < 5; ++j) {
        int jtwm1 = 2 * j;
        double absc = hlgth * XGK[jtwm1];
        double f1 = orc_townsend_pi(centr - absc, rate), f2 = orc_townsend_pi(centr + absc, rate);
        fv1[jtwm1] = f1; fv2[jtwm1] = f2;
        double fsum = f1 + f2;
        resk += WGK[jtwm1] * fsum;
        *resabs += WGK[jtwm1] * (fabs(f1) + fabs(f2));
    }
    double reskh = resk * 0.5;
    *resasc = WGK[10] * fabs(fc - reskh);
    for (int j = 0; j < 10; ++j) *resasc += WGK[j] * (fabs(fv1[j] - reskh) + fabs(fv2[j] - reskh));
    *result = resk * hlgth;
    *resabs *= dhlgth;
    *resasc *= dhlgth;
    *abserr = fabs((resk - resg) * hlgth);
    if (*resasc != 0.0 && *abserr != 0.0) {
        double q = pow(200.0 * *abserr / *resasc, 1.5);
        *abserr = *resasc * (q < 1.0 ? q : 1.0);
    }
    if (*resabs > uflow / (50.0 * epmach)) {
        double fl = (epmach * 50.0) * *resabs;
        if (fl > *abserr) *abserr = fl;
    }
}

/* ---- QUADPACK dqpsrt: keep the error list ordered ------------------------------------------- */
/* 1-based indices as in the published routine; arrays are sized limit+1 and slot 0 is unused. */
static void dqpsrt(int limit, int last, int *maxerr, double *ermax, const double *elist, int *iord,
                   int *nrmax) {
    int i, ibeg, isucc, j, jbnd, jupbn, k;
    double errmax, errmin;
    if (last <= 2) {
        iord[1] = 1; iord[2] = 2;
        goto done;
    }
    errmax = elist[*maxerr];
    if (*nrmax != 1) {
        int ido = *nrmax - 1;
        for (i = 1; i <= ido; ++i) {
            isucc = iord[*nrmax - 1];
            if (errmax <= elist[isucc]) break;
            iord[*nrmax] = isucc;
            --*nrmax;
        }
    }
    jupbn = last;
    if (last > (limit / 2 + 2)) jupbn = limit + 3 - last;
    errmin = elist[last];
    jbnd = jupbn - 1;
    ibeg = *nrmax + 1;
    if (ibeg <= jbnd) {
        for (i = ibeg; i <= jbnd; ++i) {
            isucc = iord[i];
            if (errmax >= elist[isucc]) {
                /* insert errmin by traversing the list bottom-up */
                iord[i - 1] = *maxerr;
                k = jbnd;
                for (j = i; j <= jbnd; ++j) {
                    isucc = iord[k];
                    if (errmin < elist[isucc]) {
                        iord[k + 1] = last;
                        goto done;
                    }
                    iord[k + 1] = isucc;
                    --k;
                }
                iord[i] = last;
                goto done;
            }
            iord[i - 1] = isucc;
        }
    }
    iord[jbnd] = *maxerr;
    iord[jupbn] = last;
done:
    *maxerr = iord[*nrmax];
    *ermax = elist[*maxerr];
}

/* ---- QUADPACK dqelg: epsilon algorithm ------------------------------------------------------- */
/* epstab is 1-based with 52 usable entries; res3la 1-based with 3 entries. */
static void dqelg(int *n, double *epstab, double *result, double *abserr, double *res3la, int *nres) {
    const double epmach = DBL_EPSILON, oflow = DBL_MAX;
    const int limexp = 50;
    int i, ib, ib2, ie, indx, k1, k2, k3, newelm, num;
    double delta1, delta2, delta3, e0, e1, e1abs, e2, e3, epsinf, err1, err2, err3, error, res, ss, tol1,
        tol2, tol3;
    ++*nres;
    *abserr = oflow;
    *result = epstab[*n];
    if (*n < 3) goto L100;
    epstab[*n + 2] = epstab[*n];
    newelm = (*n - 1) / 2;
    epstab[*n] = oflow;
    num = *n;
    k1 = *n;
    for (i = 1; i <= newelm; ++i) {
        k2 = k1 - 1;
        k3 = k1 - 2;
        res = epstab[k1 + 2];
        e0 = epstab[k3];
        e1 = epstab[k2];
        e2 = res;
        e1abs = fabs(e1);
        delta2 = e2 - e1;
        err2 = fabs(delta2);
        tol2 = fmax(fabs(e2), e1abs) * epmach;
        delta3 = e1 - e0;
        err3 = fabs(delta3);
        tol3 = fmax(e1abs, fabs(e0)) * epmach;
        if (err2 <= tol2 && err3 <= tol3) {
            /* e0, e1 and e2 are equal to within machine accuracy: convergence is assumed */
            *result = res;
            *abserr = err2 + err3;
            *abserr = fmax(*abserr, 5.0 * epmach * fabs(*result));
            return;
        }
        e3 = epstab[k1];
        epstab[k1] = e1;
        delta1 = e1 - e3;
        err1 = fabs(delta1);
        tol1 = fmax(e1abs, fabs(e3)) * epmach;
        /* if two elements are very close to each other, omit a part of the table */
        if (err1 <= tol1 || err2 <= tol2 || err3 <= tol3) {
            *n = i + i - 1;
            break;
        }
        ss = 1.0 / delta1 + 1.0 / delta2 - 1.0 / delta3;
        epsinf = fabs(ss * e1);
        /* test to detect irregular behaviour in the table */
        if (epsinf <= 1e-4) {
            *n = i + i - 1;
            break;
        }
        res = e1 + 1.0 / ss;
        epstab[k1] = res;
        k1 -= 2;
        error = err2 + fabs(res - e2) + err3;
        if (error <= *abserr) {
            *abserr = error;
            *result = res;
        }
    }
    /* shift the table */
    if (*n == limexp) *n = 2 * (limexp / 2) - 1;
    ib = ((num / 2) * 2 == num) ? 2 : 1;
    ie = newelm + 1;
    for (i = 1; i <= ie; ++i) {
        ib2 = ib + 2;
        epstab[ib] = epstab[ib2];
        ib = ib2;
    }
    if (num != *n) {
        indx = num - *n + 1;
        for (i = 1; i <= *n; ++i) {
            epstab[i] = epstab[indx];
            ++indx;
        }
    }
    if (*nres < 4) {
        res3la[*nres] = *result;
        *abserr = oflow;
    } else {
        *abserr = fabs(*result - res3la[3]) + fabs(*result - res3la[2]) + fabs(*result - res3la[1]);
        res3la[1] = res3la[2];
        res3la[2] = res3la[3];
        res3la[3] = *result;
    }
L100:
    *abserr = fmax(*abserr, 5.0 * epmach * fabs(*result));
}

/* ---- QUADPACK dqagse specialised to the Townsend integrand ---------------------------------- */
/* returns ier; neval_out may be NULL.  epsabs = epsrel = 1.49e-8, limit = 50 (scipy defaults). */
int orc_quad_townsend(double a, double b, double rate, double *result_out, double *abserr_out,
                      int32_t *neval_out) {
    enum { LIMIT = 50 };
    const double epsabs = 1.49e-8, epsrel = 1.49e-8;
    const double epmach = DBL_EPSILON, uflow = DBL_MIN, oflow = DBL_MAX;
    double alist[LIMIT + 2], blist[LIMIT + 2], rlist[LIMIT + 2], elist[LIMIT + 2], rlist2[53], res3la[4];
    int iord[LIMIT + 2];
    double result = 0.0, abserr = 0.0, defabs, resabs, dres, errbnd;
    int ier = 0, last, ierro = 0;
    memset(iord, 0, sizeof iord);
    alist[1] = a; blist[1] = b; rlist[1] = 0.0; elist[1] = 0.0;
    dqk21(rate, a, b, &result, &abserr, &defabs, &resabs);
    dres = fabs(result);
    errbnd = fmax(epsabs, epsrel * dres);
    last = 1;
    rlist[1] = result; elist[1] = abserr; iord[1] = 1;
    if (abserr <= 100.0 * epmach * defabs && abserr > errbnd) ier = 2;
    if (ier != 0 || (abserr <= errbnd && abserr != resabs) || abserr == 0.0) goto L140;
    {
        double errmax = abserr, area = result, errsum = abserr, small = 0.0, erlarg = 0.0, ertest = 0.0,
               correc = 0.0, erlast, reseps, abseps;
        int maxerr = 1, nrmax = 1, nres = 0, numrl2 = 2, ktmin = 0, extrap = 0, noext = 0;
        int iroff1 = 0, iroff2 = 0, iroff3 = 0, ksgn = -1, k, id, jupbnd;
        rlist2[1] = result;
        abserr = oflow;
        if (dres >= (1.0 - 50.0 * epmach) * defabs) ksgn = 1;
        for (last = 2; last <= LIMIT; ++last) {
            double a1 = alist[maxerr], b1 = 0.5 * (alist[maxerr] + blist[maxerr]), a2 = b1, b2 = blist[maxerr];
            double area1, area2, error1, error2, defab1, defab2, area12, erro12;
            erlast = errmax;
            dqk21(rate, a1, b1, &area1, &error1, &resabs, &defab1);
            dqk21(rate, a2, b2, &area2, &error2, &resabs, &defab2);
            area12 = area1 + area2;
            erro12 = error1 + error2;
            errsum = errsum + erro12 - errmax;
            area = area + area12 - rlist[maxerr];
            if (defab1 != error1 && defab2 != error2) {
                if (fabs(rlist[maxerr] - area12) <= 1e-5 * fabs(area12) && erro12 >= 0.99 * errmax) {
                    if (extrap) ++iroff2; else ++iroff1;
                }
                if (last > 10 && erro12 > errmax) ++iroff3;
            }
            rlist[maxerr] = area1;
            rlist[last] = area2;
            errbnd = fmax(epsabs, epsrel * fabs(area));
            if (iroff1 + iroff2 >= 10 || iroff3 >= 20) ier = 2;
            if (iroff2 >= 5) ierro = 3;
            if (last == LIMIT) ier = 1;
            if (fmax(fabs(a1), fabs(b2)) <= (1.0 + 100.0 * epmach) * (fabs(a2) + 1000.0 * uflow)) ier = 4;
            if (error2 > error1) {
                alist[maxerr] = a2; alist[last] = a1; blist[last] = b1;
                rlist[maxerr] = area2; rlist[last] = area1;
                elist[maxerr] = error2; elist[last] = error1;
            } else {
                alist[last] = a2; blist[maxerr] = b1; blist[last] = b2;
                elist[maxerr] = error1; elist[last] = error2;
            }
            dqpsrt(LIMIT, last, &maxerr, &errmax, elist, iord, &nrmax);
            if (errsum <= errbnd) goto L115;
            if (ier != 0) break;
            if (last == 2) {
                small = fabs(b - a) * 0.375;
                erlarg = errsum;
                ertest = errbnd;
                rlist2[2] = area;
                continue;
            }
            if (noext) continue;
            erlarg -= erlast;
            if (fabs(b1 - a1) > small) erlarg += erro12;
            if (!extrap) {
                if (fabs(blist[maxerr] - alist[maxerr]) > small) continue;
                extrap = 1;
                nrmax = 2;
            }
            if (ierro != 3 && erlarg > ertest) {
                int found = 0;
                id = nrmax;
                jupbnd = last;
                if (last > (2 + LIMIT / 2)) jupbnd = LIMIT + 3 - last;
                for (k = id; k <= jupbnd; ++k) {
                    maxerr = iord[nrmax];
                    errmax = elist[maxerr];
                    if (fabs(blist[maxerr] - alist[maxerr]) > small) { found = 1; break; }
                    ++nrmax;
                }
                if (found) continue;
            }
            /* perform extrapolation */
            ++numrl2;
            rlist2[numrl2] = area;
            dqelg(&numrl2, rlist2, &reseps, &abseps, res3la, &nres);
            ++ktmin;
            if (ktmin > 5 && abserr < 1e-3 * errsum) ier = 5;
            if (abseps < abserr) {
                ktmin = 0;
                abserr = abseps;
                result = reseps;
                correc = erlarg;
                ertest = fmax(epsabs, epsrel * fabs(reseps));
                if (abserr <= ertest) break;
            }
            if (numrl2 == 1) noext = 1;
            if (ier == 5) break;
            maxerr = iord[1];
            errmax = elist[maxerr];
            nrmax = 1;
            extrap = 0;
            small *= 0.5;
            erlarg = errsum;
        }
        if (last > LIMIT) last = LIMIT; /* loop ran to completion */
        /* L100: set final result and error estimate */
        if (abserr == oflow) goto L115;
        if (ier + ierro != 0) {
            if (ierro == 3) abserr += correc;
            if (ier == 0) ier = 3;
            if (result != 0.0 && area != 0.0) {
                if (abserr / fabs(result) > errsum / fabs(area)) goto L115;
                goto L110;
            }
            if (abserr > errsum) goto L115;
            if (area == 0.0) goto L130;
        }
    L110:
        if (ksgn == -1 && fmax(fabs(result), fabs(area)) <= defabs * 0.01) goto L130;
        if (0.01 > result / area || result / area > 100.0 || errsum > fabs(area)) ier = 6;
        goto L130;
    L115:
        result = 0.0;
        for (k = 1; k <= last; ++k) result += rlist[k];
        abserr = errsum;
    L130:
        if (ier > 2) --ier;
    }
L140:
    *result_out = result;
    *abserr_out = abserr;
    if (neval_out) *neval_out = 42 * last - 21;
    return ier;
}

/* tapir/compute.py:81-94: per interval, quad over every (finite-rate) site, then Python's
 * sequential sum() of integrals and of errors. */
void orc_net_integrals(const double *rates, int64_t n, const int32_t *intervals, int32_t n_i, int32_t mode,
                       double *sum_integral, double *sum_error) {
    for (int32_t k = 0; k < n_i; ++k) {
        double a = (double)intervals[2 * k], b = (double)intervals[2 * k + 1];
        double si = 0.0, se = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            if (!isfinite(rates[i])) continue; /* bin/tapir_compute.py:122 rates[numpy.isfinite(rates)] */
            double r, e = 0.0;
            if (mode == 0) orc_quad_townsend(a, b, rates[i], &r, &e, NULL);
            else r = orc_integral_closed(a, b, rates[i]);
            si += r; se += e;
        }
        sum_integral[k] = si;
        sum_error[k] = se;
    }
}

/* ------------------------------------------------------------------------------------------------
 * GTR eigen-system:  Q = R o pi off-diagonal, rows sum to zero (bf:978-1001).  Symmetrise with
 * S = D^{1/2} Q D^{-1/2}, D = diag(pi); Jacobi-rotate S = V L V^T; then exp(Qt) = U exp(Lt) U^-1 with
 * U = D^{-1/2} V, U^-1 = V^T D^{1/2}.
 * ---------------------------------------------------------------------------------------------- */
#define ORC_MAX_CAT 16
/* ncat > 1: discrete mixture of rate categories on top of the per-site rate (the "+G" of GTR+G; NOT in the reference's
 * script): L(s) = sum_k w_k L(s * rho_k).  ncat <= 1: the reference's model. */
typedef struct { double lam[4], U[4][4], Ui[4][4], pi[4], kappa; int ncat; double cat_rate[ORC_MAX_CAT], cat_logw[ORC_MAX_CAT]; } orc_model;

static void jacobi4(double A[4][4], double V[4][4], double w[4]) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) {
            if (A[p][q] == 0.0) continue;
            double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 4; ++k) { /* A <- A J */
                double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
            }
            for (int k = 0; k < 4; ++k) { /* A <- J^T A */
                double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
            }
            for (int k = 0; k < 4; ++k) {
                double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
            }
        }
    }
    for (int i = 0; i < 4; ++i) w[i] = A[i][i];
}

/* exch order: AC, AG, AT, CG, CT, GT (the order of the JSON "subs_matrix", bf:1024-1031) */
static void build_model(const double pi[4], const double exch[6], orc_model *m) {
    double R[4][4] = {{0, exch[0], exch[1], exch[2]}, {exch[0], 0, exch[3], exch[4]},
                      {exch[1], exch[3], 0, exch[5]}, {exch[2], exch[4], exch[5], 0}};
    double Q[4][4], S[4][4], V[4][4], sq[4];
    m->kappa = 0.0;
    for (int i = 0; i < 4; ++i) {
        double row = 0.0;
        for (int j = 0; j < 4; ++j) if (j != i) { Q[i][j] = R[i][j] * pi[j]; row += Q[i][j]; }
        Q[i][i] = -row;
        m->kappa += pi[i] * row; /* = 2 sum_{i<j} pi_i pi_j r_ij : expected substitutions per unit t*s */
        m->pi[i] = pi[i];
        sq[i] = sqrt(pi[i]);
    }
    m->ncat = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) S[i][j] = sq[i] * Q[i][j] / sq[j];
    for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) S[i][j] = S[j][i] = 0.5 * (S[i][j] + S[j][i]);
    jacobi4(S, V, m->lam);
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) {
        m->U[i][k] = V[i][k] / sq[i];
        m->Ui[k][i] = V[i][k] * sq[i];
    }
}

/* P(t) and its first two derivatives with respect to s at branch length t*s */
static void transition(const orc_model *m, double t, double s, double P[4][4], double P1[4][4], double P2[4][4]) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double p = 0, p1 = 0, p2 = 0;
        for (int k = 0; k < 4; ++k) {
            double d = m->lam[k] * t, e = exp(d * s), c = m->U[i][k] * m->Ui[k][j];
            p += c * e; p1 += c * d * e; p2 += c * d * d * e;
        }
        P[i][j] = p; P1[i][j] = p1; P2[i][j] = p2;
    }
}

typedef struct {
    int32_t nnodes, ntaxa;
    const int32_t *parent, *leaf_taxon;
    const double *blen;
    double *part; /* [nnodes][3][4] work space */
} orc_tree;

/* log L(s) and derivatives wrt u = log s for one column (pruning, bf:1053 evaluates this through
 * LikelihoodFunction siteLikelihood = (siteFilter, siteTree)). */
static void column_loglik_one(const orc_model *m, const orc_tree *tr, const uint8_t *states, int64_t ncols,
                              int64_t col, double u, double *f, double *g, double *h) {
    double s = exp(u);
    double(*X)[3][4] = (double(*)[3][4])tr->part;
    int scale = 0;
    for (int n = 0; n < tr->nnodes; ++n) {
        if (tr->leaf_taxon[n] >= 0) {
            unsigned mask = states[(int64_t)tr->leaf_taxon[n] * ncols + col] & 15u;
            if (mask == 0) mask = 15u;
            for (int i = 0; i < 4; ++i) { X[n][0][i] = (mask >> i) & 1u; X[n][1][i] = 0; X[n][2][i] = 0; }
        } else {
            for (int i = 0; i < 4; ++i) { X[n][0][i] = 1; X[n][1][i] = 0; X[n][2][i] = 0; }
        }
    }
    for (int n = 0; n < tr->nnodes; ++n) {
        int p = tr->parent[n];
        if (p < 0) continue;
        double P[4][4], P1[4][4], P2[4][4], m0[4], m1[4], m2[4];
        /* rescale an internal node whose partials got small (keeps 256+ taxa inside fp64 range) */
        if (tr->leaf_taxon[n] < 0) {
            double mx = fmax(fmax(X[n][0][0], X[n][0][1]), fmax(X[n][0][2], X[n][0][3]));
            if (mx > 0 && mx < 1e-100) {
                int e; frexp(mx, &e);
                for (int d = 0; d < 3; ++d) for (int i = 0; i < 4; ++i) X[n][d][i] = ldexp(X[n][d][i], -e);
                scale += e;
            }
        }
        transition(m, tr->blen[n], s, P, P1, P2);
        for (int i = 0; i < 4; ++i) {
            m0[i] = m1[i] = m2[i] = 0;
            for (int j = 0; j < 4; ++j) {
                m0[i] += P[i][j] * X[n][0][j];
                m1[i] += P1[i][j] * X[n][0][j] + P[i][j] * X[n][1][j];
                m2[i] += P2[i][j] * X[n][0][j] + 2 * P1[i][j] * X[n][1][j] + P[i][j] * X[n][2][j];
            }
        }
        for (int i = 0; i < 4; ++i) { /* product rule into the parent */
            double a0 = X[p][0][i], a1 = X[p][1][i], a2 = X[p][2][i];
            X[p][0][i] = a0 * m0[i];
            X[p][1][i] = a1 * m0[i] + a0 * m1[i];
            X[p][2][i] = a2 * m0[i] + 2 * a1 * m1[i] + a0 * m2[i];
        }
    }
    int root = tr->nnodes - 1;
    double L = 0, L1 = 0, L2 = 0;
    for (int i = 0; i < 4; ++i) { L += m->pi[i] * X[root][0][i]; L1 += m->pi[i] * X[root][1][i]; L2 += m->pi[i] * X[root][2][i]; }
    double gs = L1 / L, hs = L2 / L - gs * gs; /* d/ds, d2/ds2 of log L */
    *f = log(L) + scale * 0.6931471805599453;
    *g = s * gs;
    *h = s * s * hs + s * gs;
}

/* The objective the optimiser sees: one category, or the mixture sum_k w_k L(s rho_k).  With f_k, g_k, h_k the
 * log-likelihood and its u-derivatives of category k at u + log rho_k, and p_k its posterior weight:
 * f = logsumexp(log w_k + f_k), g = sum p_k g_k, h = sum p_k (h_k + g_k^2) - g^2. */
static void column_loglik(const orc_model *m, const orc_tree *tr, const uint8_t *states, int64_t ncols,
                          int64_t col, double u, double *f, double *g, double *h) {
    if (m->ncat <= 1) { column_loglik_one(m, tr, states, ncols, col, u, f, g, h); return; }
    double fk[ORC_MAX_CAT], gk[ORC_MAX_CAT], hk[ORC_MAX_CAT], top = -INFINITY;
    for (int k = 0; k < m->ncat; ++k) {
        column_loglik_one(m, tr, states, ncols, col, u + log(m->cat_rate[k]), &fk[k], &gk[k], &hk[k]);
        fk[k] += m->cat_logw[k];
        if (fk[k] > top) top = fk[k];
    }
    double z = 0, a = 0, b = 0;
    for (int k = 0; k < m->ncat; ++k) {
        const double p = exp(fk[k] - top);
        z += p; a += p * gk[k]; b += p * (hk[k] + gk[k] * gk[k]);
    }
    *f = top + log(z);
    *g = a / z;
    *h = b / z - (*g) * (*g);
}

/* Start of the search: the rate at which the tree would carry the column's Fitch parsimony count,
 * s0 = changes / (kappa * tree length * fraction of taxa present).  (HyPhy starts at siteRate = 1, bf:1050, usually
 * e^3 away from the optimum; the maximum reached is the same, in fewer evaluations.)  Sets are 4-bit masks; a node's
 * children are joined in node order. */
/* Children of every node in the order the engine's traversal joins them: deepest internal child first, tips last
 * (Sethi-Ullman order, stable among equals -- tapir_amd/csrc/tree_program.hpp).  For binary trees the order does not
 * matter to Fitch's pass; at a polytomy the sequential joins below depend on it, so the restatement uses the same one.
 * Returns kids[] (concatenated, nnodes - 1 entries) and first[] (nnodes + 1 offsets); caller frees both. */
static void fitch_child_order(const orc_tree *tr, int **kids_out, int **first_out) {
    const int nn = tr->nnodes;
    int *first = (int *)calloc((size_t)nn + 1, sizeof(int)), *kids = (int *)malloc(sizeof(int) * (size_t)(nn > 1 ? nn - 1 : 1));
    int *fill = (int *)calloc((size_t)nn, sizeof(int)), *need = (int *)calloc((size_t)nn, sizeof(int));
    for (int n = 0; n < nn; ++n) if (tr->parent[n] >= 0) ++first[tr->parent[n] + 1];
    for (int n = 0; n < nn; ++n) first[n + 1] += first[n];
    for (int n = 0; n < nn; ++n) if (tr->parent[n] >= 0) { int p = tr->parent[n]; kids[first[p] + fill[p]++] = n; }
    for (int n = 0; n < nn; ++n) { /* post-order: children before parents */
        const int a = first[n], b = first[n + 1];
        if (a == b) continue;
        for (int i = a + 1; i < b; ++i) { /* stable insertion sort by need, descending */
            const int c = kids[i];
            int j = i - 1;
            while (j >= a && need[kids[j]] < need[c]) { kids[j + 1] = kids[j]; --j; }
            kids[j + 1] = c;
        }
        int nd = 1;
        for (int i = a; i < b; ++i) {
            const int c = kids[i], extra = (i == a || need[c] == 0) ? 0 : 1;
            if (need[c] + extra > nd) nd = need[c] + extra;
        }
        need[n] = nd;
    }
    free(fill); free(need);
    *kids_out = kids; *first_out = first;
}

/* Start of the search: the rate at which the tree would carry the column's Fitch parsimony count,
 * s0 = changes / (kappa * tree length * fraction of taxa present).  (HyPhy starts at siteRate = 1, bf:1050, usually
 * e^3 away from the optimum; the maximum reached is the same, in fewer evaluations.)  Sets are 4-bit masks; a node's
 * children are joined one after the other in the order of fitch_child_order. */
static double fitch_start(const orc_model *m, const orc_tree *tr, const int *kids, const int *first, const uint8_t *states,
                          int64_t ncols, int64_t col, double chrono, int resolved, int *changes_out) {
    unsigned char *set = (unsigned char *)malloc((size_t)tr->nnodes);
    int changes = 0;
    int base_count[4] = {0, 0, 0, 0}; /* plain A / C / G / T cells of the column */
    for (int n = 0; n < tr->nnodes; ++n) { /* post-order: a node's children are complete when it is reached */
        if (tr->leaf_taxon[n] >= 0) {
            unsigned mask = states[(int64_t)tr->leaf_taxon[n] * ncols + col] & 15u;
            set[n] = (unsigned char)(mask ? mask : 15u);
            if (mask == 1u) ++base_count[0]; else if (mask == 2u) ++base_count[1]; else if (mask == 4u) ++base_count[2]; else if (mask == 8u) ++base_count[3];
            continue;
        }
        unsigned s = 15u; /* the identity of the join: all states */
        for (int i = first[n]; i < first[n + 1]; ++i) {
            const unsigned x = set[kids[i]], both = s & x;
            if (both) s = both; else { s |= x; ++changes; }
        }
        set[n] = (unsigned char)s;
    }
    free(set);
    *changes_out = changes;
    /* rate at which THIS column's states are left: sum_x p_x (-Q_xx) over the plain cells of the column, instead of the
     * stationary mean kappa = sum_x pi_x (-Q_xx).  A column of mostly fast-leaving bases reaches its parsimony count at a lower
     * site rate; with the column's own exit rate the start lands 2-3 x closer (64 taxa: rms miss 0.20 -> 0.105 log-units,
     * 256 taxa: 0.36 -> 0.13). */
    double exit_rate = m->kappa;
    {
        const int nb = base_count[0] + base_count[1] + base_count[2] + base_count[3];
        if (nb > 0) {
            double acc = 0.0;
            for (int x = 0; x < 4; ++x) {
                double qxx = 0.0;
                for (int k = 0; k < 4; ++k) qxx += m->U[x][k] * m->lam[k] * m->Ui[k][x];
                acc += (double)base_count[x] * -qxx;
            }
            if (acc > 0.0) exit_rate = acc / (double)nb;
        }
    }
    const double len = exit_rate * chrono * ((double)(resolved > 0 ? resolved : 1) / (double)tr->ntaxa);
    /* parsimony undercounts where changes are dense: stretch the count with p = changes per branch among the taxa
     * present, m' = B (-a ln(1 - p/a)), a = 0.25, p capped at 0.235 (same constants as classify_kernel; re-calibrated with
     * the column's own exit rate in the denominator: rms miss of the start 0.105 -> 0.088 log-units at 64 taxa) */
    double mch = (double)(changes > 0 ? changes : 1);
    const double B = (double)(2 * resolved - 3 > 1 ? 2 * resolved - 3 : 1);
    const double pden = fmin(mch / B, 0.235);
    mch = B * (-0.25 * log(1.0 - pden / 0.25));
    double u0 = (len > 0) ? log(mch / len) : 0.0;
    if (!(u0 == u0)) u0 = 0.0;
    if (u0 < -20.0) u0 = -20.0;
    if (u0 > 8.0) u0 = 8.0;
    return u0;
}

/* Safeguarded Newton on u = log s from u_start, to the local maximum uphill of the start. */
/* mfitch: the column's parsimony length when the start is the parsimony start (used by the first step), else 0 */
static void maximise_column(const orc_model *m, const orc_tree *tr, const uint8_t *states, int64_t ncols, int64_t col,
                            double u_start, double mfitch, double *s_out, double *f_out, uint8_t *flag_out, int32_t *neval) {
    double u = u_start, lo = ORC_U_MIN, hi = ORC_U_MAX, f = 0, g, h;
    double u_prev = 0, h_prev = 0, g_prev = 0, f_prev = 0;
    int lo_open = 1, hi_open = 1, have_prev = 0; /* bracket ends not evaluated yet; previous point known */
    *flag_out = 4;
    for (int it = 0; it < ORC_MAXIT; ++it) {
        column_loglik(m, tr, states, ncols, col, u, &f, &g, &h);
        ++*neval;
#ifdef ORC_TRACE
        fprintf(stderr, "  eval %d: u %.12f f %.12f g %.6e h %.6e\n", it, u, f, g, h);
#endif
        int uphill = !(g <= 0); /* NaN (L underflowed to 0 at tiny s) counts as uphill */
        /* Saturation: log L has reached its s -> infinity asymptote to within fp64 resolution.  Beyond this
         * point g is a second-order-small number buried under first-order rounding noise (its sign is
         * meaningless), so the policy value s = 1e4 is reported instead of chasing it. */
        if (fabs(g) < ORC_FLAT_EPS && fabs(h) < ORC_FLAT_EPS) { *flag_out = 2; u = ORC_U_MAX; break; }
        if (u >= ORC_U_MAX && uphill) { *flag_out = 2; break; }
        if (u <= ORC_U_MIN && !uphill) { *flag_out = 3; break; }
        if (uphill) { lo = u; lo_open = 0; } else { hi = u; hi_open = 0; }
        /* Step: where f is concave, the Newton step -g/h refined to log(1 - g/h), which is the exact maximiser of
         * the model f(u) = m u - a exp(u) + c fitted to (g, h) -- the generic shape of a site log-likelihood in
         * u = log(rate) (m ~ number of substitutions, a exp(u) ~ expected number).  It agrees with Newton to
         * second order in the step (so convergence stays quadratic) and saves about one evaluation in five from
         * the far starting point u = 0.  Where f is convex: a capped step uphill. */
        double step;
        if (h < 0) {
            const double q = 1.0 - g / h;
            step = (q > 0) ? log(q) : -g / h;
            /* First step from the parsimony start: as u -> -inf the slope of log L tends to the column's parsimony length m.
             * Fitting m u - a exp(b u) + c to (m, g, h) -- one more shape parameter than the model above, whose m is implied
             * by g - h -- gives (1/b) log(m / (m - g)), b = -h / (m - g): nearer the optimum than the step above on 99 % of
             * columns (tools/debug/step_rule_experiment.py), 3-5 % fewer evaluations; same rule as site_rate_kernel. */
            if (!have_prev && mfitch > 0.0) {
                const double A = mfitch - g;
                const double sb = (A > 0.0) ? log(mfitch / A) * (A / -h) : ORC_STEP_MAX + 1.0;
                if (fabs(sb) <= ORC_STEP_MAX) step = sb;
            }
        } else {
            step = uphill ? ORC_STEP_MAX : -ORC_STEP_MAX;
        }
        /* Weakly curved concave points (|h| below ORC_HERMITE_REGULAR) with a previous concave point nearby: the step of the
         * model above assumes f''' = f'', which is what a column looks like from far below its optimum; on the approach to a
         * flat maximum at a large rate f'' decays like f' instead, Newton-type steps then converge linearly (eight steps of
         * 0.07 ... 0.002 were seen), and in a small batch one such lane keeps its wave and the launch alive.  Halley's step
         * with the third derivative measured from the last two curvatures, -g / (h - g f3 / (2 h)), converges cubically there:
         * on 5-taxon columns the slowest of 24 000 needs 17 evaluations instead of 29 and 0.24 % instead of 1.3 % need 14 or
         * more; trees of 64 and 256 taxa are untouched (same rule as site_rate_kernel). */
        if (have_prev && h < 0 && h_prev < 0 && fabs(h) < ORC_HERMITE_REGULAR && fabs(u - u_prev) < ORC_HALLEY_SPAN) {
            const double f3 = (h - h_prev) / (u - u_prev);
            const double den = h - 0.5 * g * f3 / h;
            if (den < 0) {
                const double sh = -g / den;
                if (fabs(sh) <= ORC_STEP_MAX && (sh > 0) == (g > 0)) step = sh;
            }
        }
        if (!(step <= ORC_STEP_MAX)) step = ORC_STEP_MAX;
        if (step < -ORC_STEP_MAX) step = -ORC_STEP_MAX;
        /* Plateau stride: still uphill at a rate of 20 or more with nothing known above is almost always a column whose
         * log L creeps up to its s -> infinity asymptote; g and h shrink together there, the model behind the step sees a
         * maximum just ahead and the steps stay at ~0.1 for 20-30 evaluations until the flatness rule fires.  A stride
         * of at least ORC_PLATEAU_STRIDE covers the two log-units to flatness in a few evaluations; a maximum that does
         * lie ahead is overshot by at most that much and then bracketed from both sides. */
        if (uphill && hi_open && u >= ORC_U_CHECK && step < ORC_PLATEAU_STRIDE) step = ORC_PLATEAU_STRIDE;
        /* With two evaluations in hand, f' is known with its slope at both points, and so is its integral between them
         * (the difference of the two values): the quartic through those five conditions locates the zero of f' to fifth
         * order, so a remaining step of up to ORC_HERMITE_TOL can be taken WITHOUT evaluating again.  (Round 1 used the
         * cubic through the four slope conditions: ~15 x the error at the same step, hence bounds of 1e-3 / 2e-4 and 60 % of
         * the columns leaving after two evaluations; DESIGN section 8 r2 has the calibration of the present bounds.)
         * Both exits are for REGULAR points only -- curvature of at least ORC_HERMITE_REGULAR, i.e. a tenth of a
         * substitution's worth (|h| ~ the number of changes at the optimum of a regular column).  A weakly curved column
         * sits near an inflection or on the approach to a plateau (|h| ~ 1e-3 and changing by 20 % per 0.01 log-units);
         * no low-order model of f' holds there -- round 1's exits left 2e-7 ... 1.4e-6 on one such column in 3 x 10^5 --
         * so it iterates until the step itself is below ORC_STEP_TOL_FIRST. */
        const int regular_here = fabs(h) >= ORC_HERMITE_REGULAR;
        if (have_prev && h < 0 && regular_here && fabs(h_prev) >= ORC_HERMITE_REGULAR) {
            const double d = u - u_prev; /* the previous point sits at t = -d */
            const double r1 = g_prev - g + h * d, r2 = (h_prev - h) * d, r3 = (f - f_prev) / d - g + 0.5 * h * d;
            const double id = 1.0 / d, id2 = id * id;
            const double q2 = (-12.0 * r1 - 1.5 * r2 + 30.0 * r3) * id2;
            const double q3 = (-28.0 * r1 - 4.0 * r2 + 60.0 * r3) * (id2 * id);
            const double q4 = (-15.0 * r1 - 2.5 * r2 + 30.0 * r3) * (id2 * id2);
            double t = -g / h;
            for (int k = 0; k < 4; ++k) { /* Newton on g + h t + q2 t^2 + q3 t^3 + q4 t^4 */
                const double p = g + t * (h + t * (q2 + t * (q3 + t * q4)));
                const double dp = h + t * (2.0 * q2 + t * (3.0 * q3 + 4.0 * t * q4));
                if (dp < 0) t -= p / dp;
            }
            /* The quartic is the derivative of the quintic Hermite interpolant of f (three conditions at each point), whose
             * error is f^(6) / 6! t^3 (t + d)^3: the zero of f' is off by K t^2 |d|^3 with K = |f^(6)| / (240 |h|), hence a
             * bound on t^2 |d|^3 itself.  K is ~0.004 on the median column, ~15 on fast sites whose series in u has a radius
             * of ~0.4 -- and reaches 300 on the wildest of C4's and C5's 57 M optimised columns (curvature halving within 0.1
             * log-units: one in 10^7), where only a small step helps: the error is what the model gets wrong in the third
             * derivative times t^2 / |h|, up to ~t^2 / 30.  Wider bounds were measured (step 1e-2, span 1e-3: 87 % of the
             * columns leave after two evaluations instead of 67 %) and left 1.1e-6 on one column of C5; an agreement test
             * between the quartic and the cubic does not catch those columns -- both interpolants are off by the same amount.
             * Further: the far point must not vouch for a step it is too far from (span), the higher-order terms must be a
             * small correction of h, and the difference of two rounded values must not steer the zero (the rounding of f
             * reaches q2 as ~30 eps |f| / |d|^3 and moves the zero by that times t^2 / |h|, kept below 1e-9). */
            if (fabs(t) < ORC_HERMITE_TOL && fabs(t) < 0.5 * fabs(d) && fabs(t * d) < ORC_HERMITE_SPAN &&
                t * t * fabs(d * d * d) < ORC_HERMITE_T2D3 &&
                fabs(t * (q2 + t * (q3 + t * q4))) < ORC_HERMITE_GUARD * fabs(h) &&
                ORC_HERMITE_NOISE * fabs(f) * (t * t) < fabs(h) * fabs(d * d * d)) {
                f += t * (g + t * (0.5 * h + t * (q2 / 3.0 + t * (0.25 * q3 + t * (0.2 * q4)))));
                u += t;
                *flag_out = 0;
                if (u >= ORC_U_CHECK) { /* same confirmation by value as below */
                    double fm, gm, hm;
                    column_loglik(m, tr, states, ncols, col, ORC_U_MAX, &fm, &gm, &hm);
                    ++*neval;
                    if (fm >= f - ORC_SAT_TOL * fmax(1.0, fabs(f))) { *flag_out = 2; u = ORC_U_MAX; f = fm; }
                }
                break;
            }
        }
        const double tol = (have_prev && regular_here) ? ORC_STEP_TOL : ORC_STEP_TOL_FIRST;
        double un = u + step;
        /* the bracket safeguard must not see a converged (possibly underflowing) step */
        if (fabs(step) >= tol) {
            if (un >= hi) un = hi_open ? ORC_U_MAX : 0.5 * (lo + hi);
            else if (un <= lo) un = lo_open ? ORC_U_MIN : 0.5 * (lo + hi);
            step = un - u;
        }
        /* Converged?  Rather than spend one more evaluation to watch the step shrink from ~1e-4 to ~1e-8, the last
         * step gets its third-order correction: the third derivative f3 from the two most recent curvatures (the
         * step's own model already carries f3 = h), leaving an error of O(step^3), i.e. ~1e-8 or less in u.  Where that
         * correction is a sizeable part of the step itself (weak curvature, far previous point) the model behind it is
         * not good enough to stop on: iterate once more unless the step is already negligible. */
        double f3 = h, corr = 0.0;
        if (have_prev && h < 0) {
            f3 = (h - h_prev) / (u - u_prev);
            corr = 0.5 * ((f3 - h) / h) * step * step;
        }
        if (fabs(step) < tol && (fabs(corr) <= 0.05 * fabs(step) || fabs(step) < ORC_STEP_TOL_FIRST)) {
            step -= corr;
            f += step * (g + step * (0.5 * h + step * (f3 / 6.0)));
            u += step;
            *flag_out = 0;
            /* A maximum this far out may be rounding noise on the plateau log L reaches as s -> infinity (where the
             * flatness rule fires or not depending on the last bits): confirm it by value against the largest rate. */
            if (u >= ORC_U_CHECK) {
                double fm, gm, hm;
                column_loglik(m, tr, states, ncols, col, ORC_U_MAX, &fm, &gm, &hm);
                ++*neval;
                if (fm >= f - ORC_SAT_TOL * fmax(1.0, fabs(f))) { *flag_out = 2; u = ORC_U_MAX; f = fm; }
            }
            break;
        }
        u_prev = u; h_prev = h; g_prev = g; f_prev = f; have_prev = 1;
        u = un;
    }
    *s_out = exp(u);
    *f_out = f;
}

/* REFERENCE-FAITHFUL MODE (start_mode = 1): what the script literally asks HyPhy for -- every column starts at
 * siteRate = 1 (bf:1050) and Optimize() climbs from there (bf:1053) -- restated with nothing borrowed from the product's
 * accelerated optimiser: start u = log 1 = 0, plain Newton step -f'/f'' where f is concave (a capped step uphill where
 * it is not), bracket safeguard, iterate until the step is below 1e-12.  No parsimony start, no log-step refinement, no
 * plateau stride, no Hermite / third-order exits.  Only the reporting POLICY is shared with the default mode (DESIGN.md
 * section 5: when a column counts as saturated, the s = 1e4 policy value, the confirmation by value beyond s = 20), so
 * that the two modes are comparable column by column: a difference in `rate` between them is a different local optimum
 * (or an inaccurate accelerated exit), nothing else.  SURVEY F4: HyPhy returns the local optimum uphill of its start. */
static void maximise_column_plain(const orc_model *m, const orc_tree *tr, const uint8_t *states, int64_t ncols, int64_t col,
                                  double *s_out, double *f_out, uint8_t *flag_out, int32_t *neval) {
    double u = 0.0, lo = ORC_U_MIN, hi = ORC_U_MAX, f = 0, g, h;
    int lo_open = 1, hi_open = 1;
    *flag_out = 4;
    for (int it = 0; it < 4 * ORC_MAXIT; ++it) {
        column_loglik(m, tr, states, ncols, col, u, &f, &g, &h);
        ++*neval;
        int uphill = !(g <= 0);
        if (fabs(g) < ORC_FLAT_EPS && fabs(h) < ORC_FLAT_EPS) { *flag_out = 2; u = ORC_U_MAX; break; }
        if (u >= ORC_U_MAX && uphill) { *flag_out = 2; break; }
        if (u <= ORC_U_MIN && !uphill) { *flag_out = 3; break; }
        if (uphill) { lo = u; lo_open = 0; } else { hi = u; hi_open = 0; }
        double step = (h < 0) ? -g / h : (uphill ? ORC_STEP_MAX : -ORC_STEP_MAX);
        if (!(step <= ORC_STEP_MAX)) step = ORC_STEP_MAX;
        if (step < -ORC_STEP_MAX) step = -ORC_STEP_MAX;
        double un = u + step;
        if (fabs(step) >= 1e-12) {
            if (un >= hi) un = hi_open ? ORC_U_MAX : 0.5 * (lo + hi);
            else if (un <= lo) un = lo_open ? ORC_U_MIN : 0.5 * (lo + hi);
            /* a bracket that has closed to rounding is convergence too */
            if (!(hi_open || lo_open) && hi - lo < 1e-12) { step = 0.0; un = u; }
            else { u = un; continue; }
        }
        u = un;
        *flag_out = 0;
        if (u >= ORC_U_CHECK) {
            double fm, gm, hm;
            column_loglik(m, tr, states, ncols, col, ORC_U_MAX, &fm, &gm, &hm);
            ++*neval;
            if (fm >= f - ORC_SAT_TOL * fmax(1.0, fabs(f))) { *flag_out = 2; u = ORC_U_MAX; f = fm; }
        }
        break;
    }
    *s_out = exp(u);
    *f_out = f;
}

/* One locus.  Outputs per column: rate = kappa*s (bf:1061), subst = rate*chronoLength (bf:1056-1060),
 * lnl, flag, nres = number of plain A/C/G/T cells (tapir/compute.py:104). Returns total evaluations. */
static int64_t site_rates_impl(const uint8_t *states, int64_t ncols, int32_t ntaxa, int32_t nnodes, const int32_t *parent,
                               const double *blen, const int32_t *leaf_taxon, const double *pi, const double *exch,
                               int32_t ncat, const double *cat_rate, const double *cat_weight, int32_t start_mode,
                               double *rate, double *subst, double *lnl, uint8_t *flag, int32_t *nres_out) {
    orc_model m;
    orc_tree tr = {nnodes, ntaxa, parent, leaf_taxon, blen, NULL};
    int64_t total_eval = 0;
    double chrono = 0.0;
    build_model(pi, exch, &m);
    if (ncat > 1 && ncat <= ORC_MAX_CAT) {
        m.ncat = ncat;
        for (int k = 0; k < ncat; ++k) { m.cat_rate[k] = cat_rate[k]; m.cat_logw[k] = log(cat_weight[k]); }
    }
    for (int n = 0; n < nnodes; ++n) if (parent[n] >= 0) chrono += blen[n]; /* bf:1006-1013 */
    tr.part = (double *)malloc(sizeof(double) * 12 * (size_t)nnodes);
    int *kids = NULL, *first = NULL;
    fitch_child_order(&tr, &kids, &first);
    for (int64_t c = 0; c < ncols; ++c) {
        unsigned uni = 0; int informative = 0, resolved = 0;
        for (int n = 0; n < nnodes; ++n) {
            if (leaf_taxon[n] < 0) continue;
            unsigned mask = states[(int64_t)leaf_taxon[n] * ncols + c] & 15u;
            if (mask == 0) mask = 15u;
            if (mask != 15u) { uni |= mask; ++resolved; }
            if (mask == 1u || mask == 2u || mask == 4u || mask == 8u) ++informative;
        }
        nres_out[c] = informative;
        double s, f; uint8_t fl; int32_t ne = 0;
        if (resolved <= 1) {
            /* flat: L(s) = pi_x (or 1): independent of s; siteRate keeps its start value 1 (bf:1050) */
            double g, h; column_loglik(&m, &tr, states, ncols, c, 0.0, &f, &g, &h); ne = 1;
            s = 1.0; fl = 1;
        } else if ((uni & (uni - 1)) == 0) {
            /* every resolved taxon carries the same base x: L(s) <= pi_x = L(0), optimum at s = 0 */
            int x = (uni == 1) ? 0 : (uni == 2) ? 1 : (uni == 4) ? 2 : 3;
            s = 0.0; f = log(m.pi[x]); fl = 3;
        } else if (start_mode == 1) {
            maximise_column_plain(&m, &tr, states, ncols, c, &s, &f, &fl, &ne);
        } else if (start_mode == 2) { /* HyPhy's start value, the product's step rule and exits (tphip_plan_desc.start_rule = 1) */
            maximise_column(&m, &tr, states, ncols, c, 0.0, 0.0, &s, &f, &fl, &ne);
        } else {
            int changes = 0;
            const double u0 = fitch_start(&m, &tr, kids, first, states, ncols, c, chrono, resolved, &changes);
            maximise_column(&m, &tr, states, ncols, c, u0, ntaxa >= ORC_FIRST_STEP_MIN_TAXA ? (double)changes : 0.0, &s, &f, &fl, &ne);
        }
        total_eval += ne;
        rate[c] = s * m.kappa;
        subst[c] = rate[c] * chrono;
        lnl[c] = f;
        flag[c] = fl;
    }
    free(tr.part);
    free(kids); free(first);
    return total_eval;
}

int64_t orc_site_rates_mix(const uint8_t *states, int64_t ncols, int32_t ntaxa, int32_t nnodes, const int32_t *parent,
                           const double *blen, const int32_t *leaf_taxon, const double *pi, const double *exch,
                           int32_t ncat, const double *cat_rate, const double *cat_weight,
                           double *rate, double *subst, double *lnl, uint8_t *flag, int32_t *nres_out) {
    return site_rates_impl(states, ncols, ntaxa, nnodes, parent, blen, leaf_taxon, pi, exch, ncat, cat_rate, cat_weight, 0, rate,
                           subst, lnl, flag, nres_out);
}

int64_t orc_site_rates(const uint8_t *states, int64_t ncols, int32_t ntaxa, int32_t nnodes, const int32_t *parent,
                       const double *blen, const int32_t *leaf_taxon, const double *pi, const double *exch,
                       double *rate, double *subst, double *lnl, uint8_t *flag, int32_t *nres_out) {
    return site_rates_impl(states, ncols, ntaxa, nnodes, parent, blen, leaf_taxon, pi, exch, 0, NULL, NULL, 0, rate, subst, lnl,
                           flag, nres_out);
}

/* start_mode 0: the product's optimiser (parsimony start, accelerated exits); 1: the reference-faithful restatement
 * (start at siteRate = 1, plain Newton to 1e-12; see maximise_column_plain); 2: the product's optimiser started at
 * siteRate = 1 (what the engine runs with start_rule = 1) */
int64_t orc_site_rates_mode(const uint8_t *states, int64_t ncols, int32_t ntaxa, int32_t nnodes, const int32_t *parent,
                            const double *blen, const int32_t *leaf_taxon, const double *pi, const double *exch,
                            int32_t start_mode, double *rate, double *subst, double *lnl, uint8_t *flag, int32_t *nres_out) {
    return site_rates_impl(states, ncols, ntaxa, nnodes, parent, blen, leaf_taxon, pi, exch, 0, NULL, NULL, start_mode, rate,
                           subst, lnl, flag, nres_out);
}

/* log-likelihood curve of one column at given u values (used by tests to check derivatives) */
void orc_column_curve(const uint8_t *states, int64_t ncols, int32_t ntaxa, int32_t nnodes, const int32_t *parent,
                      const double *blen, const int32_t *leaf_taxon, const double *pi, const double *exch, int64_t col,
                      const double *u, int32_t nu, double *f, double *g, double *h) {
    orc_model m;
    orc_tree tr = {nnodes, ntaxa, parent, leaf_taxon, blen, NULL};
    build_model(pi, exch, &m);
    tr.part = (double *)malloc(sizeof(double) * 12 * (size_t)nnodes);
    for (int i = 0; i < nu; ++i) column_loglik(&m, &tr, states, ncols, col, u[i], &f[i], &g[i], &h[i]);
    free(tr.part);
}

/* eigen-system export for tests: lam[4], U[16], Ui[16], kappa */
void orc_gtr_eigen(const double *pi, const double *exch, double *lam, double *U, double *Ui, double *kappa) {
    orc_model m;
    build_model(pi, exch, &m);
    memcpy(lam, m.lam, sizeof m.lam);
    memcpy(U, m.U, sizeof m.U);
    memcpy(Ui, m.Ui, sizeof m.Ui);
    *kappa = m.kappa;
}

/* tapir/compute.py:96-106: per column count of cells whose character is one of A,T,G,C */
void orc_informative_counts(const uint8_t *states, int64_t ncols, int32_t ntaxa, int32_t *counts) {
    for (int64_t c = 0; c < ncols; ++c) {
        int k = 0;
        for (int t = 0; t < ntaxa; ++t) {
            unsigned mask = states[(int64_t)t * ncols + c] & 15u;
            k += (mask == 1u || mask == 2u || mask == 4u || mask == 8u);
        }
        counts[c] = k;
    }
}

/* HyPhy stage 1 objective (bf:487-520, 647-655): sum over columns of log L with every site at rate 1,
 * for given exchangeabilities and branch lengths (plain post-order recursion, one column at a time). */
double orc_locus_loglik(const uint8_t *states, int64_t ncols, int32_t ntaxa, int32_t nnodes, const int32_t *parent,
                        const double *blen, const int32_t *leaf_taxon, const double *pi, const double *exch) {
    orc_model m;
    orc_tree tr = {nnodes, ntaxa, parent, leaf_taxon, blen, NULL};
    double total = 0.0, f, g, h;
    build_model(pi, exch, &m);
    tr.part = (double *)malloc(sizeof(double) * 12 * (size_t)nnodes);
    for (int64_t c = 0; c < ncols; ++c) {
        column_loglik(&m, &tr, states, ncols, c, 0.0, &f, &g, &h);
        total += f;
    }
    free(tr.part);
    return total;
}
