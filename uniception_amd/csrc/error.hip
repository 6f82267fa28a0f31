// Thread-local error text, ABI version, build flavour and the tuning knobs of libuc_hip.so.
#include "common.h"
#include "knobs.h"

#include <mutex>
#include <stdlib.h>
#include <string.h>

static thread_local char g_uc_err[512] = {0};

void uc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_uc_err, sizeof(g_uc_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* uc_last_error(void) { return g_uc_err; }
extern "C" int uc_abi_version(void) { return UC_ABI_VERSION; }

#ifdef UC_DIAG
extern "C" const char* uc_build_flavor(void) { return "diag"; }
#else
extern "C" const char* uc_build_flavor(void) { return "release"; }
#endif

#define UC_ATTN_RS_DEFAULT 0
static UcKnobs g_knobs;
static std::once_flag g_knobs_once;
std::atomic<int> g_uc_gemm_variant{-3};
std::atomic<int> g_uc_gemm_stagger{-1};
std::atomic<int> g_uc_attn_rs{UC_ATTN_RS_DEFAULT};
std::atomic<int> g_uc_attn_p64{1};
std::atomic<int> g_uc_attn_bwd64{1};
std::atomic<int> g_uc_conv_rows{1};
std::atomic<int> g_uc_conv_rows_flat{1};
std::atomic<int> g_uc_small_m_split{2048};

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

const UcKnobs& uc_knobs() {
    std::call_once(g_knobs_once, [] {
        g_knobs.gemm_group_m = env_int("UC_GEMM_GROUP_M", 4);
        if (g_knobs.gemm_group_m < 1) g_knobs.gemm_group_m = 1;
        g_knobs.gemm_coresident = env_int("UC_GEMM_CORESIDENT", 1);
        g_knobs.gemm_nt = env_int("UC_GEMM_NT", -1);
        g_knobs.gemm_8wave = env_int("UC_GEMM_8WAVE", 1);
        g_knobs.gemm_small_stages = env_int("UC_GEMM_SMALL_STAGES", 3);
        g_knobs.gemm_4wave = env_int("UC_GEMM_4WAVE", 3);
        g_knobs.gemm_4wave_min_k = env_int("UC_GEMM_4WAVE_MIN_K", 2048);
        g_knobs.gemm_side_lds = env_int("UC_GEMM_SIDE_LDS", 1);
        g_knobs.conv_dw_rows = env_int("UC_CONV_DW_ROWS", 1);
        g_knobs.attn_nw = env_int("UC_ATTN_NW", 0);
        g_knobs.attn_dma = env_int("UC_ATTN_DMA", 1);
        g_knobs.attn_prio = env_int("UC_ATTN_PRIO", 0);
        g_knobs.bilinear_rows2 = env_int("UC_BILINEAR_ROWS2", 4);
        g_knobs.ln_nt = env_int("UC_LN_NT", -1);
#ifdef UC_DIAG
        g_knobs.gemm_dbg = env_int("UC_GEMM_DBG", 0);
        g_knobs.attn_dbg = env_int("UC_ATTN_DBG", 0);
        g_knobs.gemm_trace = env_int("UC_GEMM_TRACE", 0);
#endif
        g_uc_gemm_variant.store(env_int("UC_GEMM_VARIANT", -3));
        g_uc_gemm_stagger.store(env_int("UC_GEMM_STAGGER", -1));
        g_uc_attn_rs.store(env_int("UC_ATTN_RS", UC_ATTN_RS_DEFAULT));
        g_uc_attn_p64.store(env_int("UC_ATTN_P64", 1));
        g_uc_attn_bwd64.store(env_int("UC_ATTN_BWD64", 1));
        g_uc_conv_rows.store(env_int("UC_CONV_ROWS", 1));
        g_uc_conv_rows_flat.store(env_int("UC_CONV_ROWS_FLAT", 1));
        g_uc_small_m_split.store(env_int("UC_GEMM_SMALLM", 2048));
    });
    return g_knobs;
}

extern "C" int uc_tuning_set(const char* name, int value) {
    UC_REQUIRE(name, "uc_tuning_set: null name");
    (void)uc_knobs();   // the environment's initial values first, so that a later first use does not overwrite this call
    if (!strcmp(name, "gemm_variant")) {
        UC_REQUIRE(value == -3 || value == -1 || (value >= 0 && value <= 4) || value == 6 || value == 7, "uc_tuning_set: gemm_variant must be -3 (automatic), -1, 0..4, 6 or 7 (got %d)", value);
        g_uc_gemm_variant.store(value);
    } else if (!strcmp(name, "gemm_stagger")) {
        UC_REQUIRE(value >= -1 && value <= 100000, "uc_tuning_set: gemm_stagger out of range (%d)", value);
        g_uc_gemm_stagger.store(value);
    } else if (!strcmp(name, "attn_role_split")) {
        UC_REQUIRE(value == 0 || value == 1, "uc_tuning_set: attn_role_split must be 0 or 1 (got %d)", value);
        g_uc_attn_rs.store(value);
    } else if (!strcmp(name, "attn_p64")) {
        UC_REQUIRE(value >= 0 && value <= 2, "uc_tuning_set: attn_p64 must be 0 (never), 1 (policy) or 2 (wherever the shape allows) (got %d)", value);
        g_uc_attn_p64.store(value);
    } else if (!strcmp(name, "attn_bwd64")) {
        UC_REQUIRE(value >= 0 && value <= 2, "uc_tuning_set: attn_bwd64 must be 0 (never), 1 (policy) or 2 (always) (got %d)", value);
        g_uc_attn_bwd64.store(value);
    } else if (!strcmp(name, "small_m_split")) {
        UC_REQUIRE(value >= 0, "uc_tuning_set: small_m_split is the smallest K a small-M launch splits in two for (0: never) (got %d)", value);
        g_uc_small_m_split.store(value);
    } else if (!strcmp(name, "conv_rows")) {
        UC_REQUIRE(value >= 0 && value <= 3, "uc_tuning_set: conv_rows must be 0 (implicit GEMM everywhere), 1 (row-walking kernels where they win), 2 (the 256-pixel row-walking kernel wherever the shape allows) or 3 (the eight-wave 512-pixel one wherever the shape allows) (got %d)", value);
        g_uc_conv_rows.store(value);
    } else if (!strcmp(name, "conv_rows_flat")) {
        UC_REQUIRE(value == 0 || value == 1, "uc_tuning_set: conv_rows_flat must be 0 (the eight-wave row kernel only on maps whose rows tile 512 pixels) or 1 (its flat form on every other map too) (got %d)", value);
        g_uc_conv_rows_flat.store(value);
    } else {
        uc_set_error("uc_tuning_set: unknown knob '%s' (run-time switchable: gemm_variant, gemm_stagger, attn_role_split, attn_p64, attn_bwd64, conv_rows, conv_rows_flat, small_m_split; everything else is read from the environment once, see csrc/knobs.h)", name);
        return UC_ERR_BAD_ARG;
    }
    return UC_OK;
}

extern "C" int uc_tuning_get(const char* name, int* value) {
    UC_REQUIRE(name && value, "uc_tuning_get: null argument");
    (void)uc_knobs();
    if (!strcmp(name, "gemm_variant")) *value = g_uc_gemm_variant.load();
    else if (!strcmp(name, "gemm_stagger")) *value = g_uc_gemm_stagger.load();
    else if (!strcmp(name, "attn_role_split")) *value = g_uc_attn_rs.load();
    else if (!strcmp(name, "attn_p64")) *value = g_uc_attn_p64.load();
    else if (!strcmp(name, "attn_bwd64")) *value = g_uc_attn_bwd64.load();
    else if (!strcmp(name, "conv_rows")) *value = g_uc_conv_rows.load();
    else if (!strcmp(name, "conv_rows_flat")) *value = g_uc_conv_rows_flat.load();
    else if (!strcmp(name, "small_m_split")) *value = g_uc_small_m_split.load();
    else { uc_set_error("uc_tuning_get: unknown knob '%s'", name); return UC_ERR_BAD_ARG; }
    return UC_OK;
}
