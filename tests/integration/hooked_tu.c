/* tests/integration/hooked_tu.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Builds oracle/_ref/libsecp256k1_hooked.so: the unmodified reference translation unit (through oracle/ref_shim.c, which
 * #includes src/secp256k1.c from where it lies) + integration/secp256k1_amd_hook.c, wired the way a maintainer would wire
 * it: every call site of the static secp256k1_ecmult_multi_var *after* its own definition -- i.e. the modules (BP++ norm
 * argument, bppp_norm_product_impl.h:386,397,543) and the shim's ref_ecmult_multi -- goes through the adapter
 * secp256k1_ecmult_multi_var_amd.  The redirect is a one-line macro placed after ecmult_impl.h has been included under its
 * real name (same include prefix as src/secp256k1.c:18-31), so no reference source is edited or copied.
 */
#define SECP256K1_BUILD
#include "../include/secp256k1.h"
#include "../include/secp256k1_preallocated.h"
#include "assumptions.h"
#include "checkmem.h"
#include "util.h"
#include "field_impl.h"
#include "scalar_impl.h"
#include "group_impl.h"
#include "ecmult_impl.h"

static int secp256k1_ecmult_multi_var_amd(const secp256k1_callback *error_callback, secp256k1_scratch *scratch, secp256k1_gej *r,
        const secp256k1_scalar *inp_g_sc, secp256k1_ecmult_multi_callback cb, void *cbdata, size_t n);
#define secp256k1_ecmult_multi_var secp256k1_ecmult_multi_var_amd
#include "ref_shim.c"
#undef secp256k1_ecmult_multi_var

#include "secp256k1_amd_hook.c"

/* ---- drivers for tests/test_cpu_hook.py ---- */
typedef struct { const unsigned char *sc; const unsigned char *pt; const unsigned char *inf; long fail_at; size_t calls; } hook_cbdata;
static int hook_cb(secp256k1_scalar *sc, secp256k1_ge *pt, size_t idx, void *data) {
    hook_cbdata *d = (hook_cbdata *)data;
    d->calls++;
    if (d->fail_at >= 0 && (size_t)d->fail_at == idx) return 0;
    ref_scalar_from_b32(sc, d->sc + 32 * idx);
    ref_ge_from_b64(pt, d->pt + 64 * idx, d->inf ? d->inf[idx] : 0);
    return 1;
}
/* returns -1 when the adapter returned 0, else the infinity flag; *calls = how often the callback ran */
REF_EXPORT int hook_test_ecmult_multi(unsigned char *r64, const unsigned char *g_sc32, const unsigned char *sc32, const unsigned char *pt64,
                                      const unsigned char *inf, size_t n, long fail_at, size_t *calls) {
    secp256k1_context *ctx = secp256k1_context_create(SECP256K1_CONTEXT_NONE);
    secp256k1_scratch *scratch = secp256k1_scratch_space_create(ctx, 64 * 1000 * 1000);
    secp256k1_scalar g; secp256k1_gej rj; hook_cbdata d; int ok;
    d.sc = sc32; d.pt = pt64; d.inf = inf; d.fail_at = fail_at; d.calls = 0;
    if (g_sc32) ref_scalar_from_b32(&g, g_sc32);
    ok = secp256k1_ecmult_multi_var_amd(&ctx->error_callback, scratch, &rj, g_sc32 ? &g : NULL, hook_cb, &d, n);
    if (calls) *calls = d.calls;
    secp256k1_scratch_space_destroy(ctx, scratch);
    secp256k1_context_destroy(ctx);
    if (!ok) return -1;
    return ref_gej_to_b64(r64, &rj);
}
