// options.hip — the library's option table (options.hpp) and its C-ABI face: nfa_set_option / nfa_get_option /
// nfa_reset_options.  Host code only.
#include <cctype>
#include <cstring>
#include <mutex>
#include <string>

#include "common.hpp"
#include "options.hpp"

namespace nfa {
namespace {

struct OptionSpec {
    const char *name;                 // canonical lower-case name; "NFA_<NAME>" is the environment variable that seeds it
    const char *doc;
    bool (*parse)(const char *, int64_t *);
};

bool parse_int(const char *s, int64_t *out) {
    char *end = nullptr;
    const long long v = strtoll(s, &end, 10);
    if (end == s || *end != '\0') return false;
    *out = v;
    return true;
}
template <int64_t... ALLOWED>
bool parse_one_of(const char *s, int64_t *out) {
    int64_t v;
    if (!parse_int(s, &v)) return false;
    const int64_t allowed[] = {ALLOWED...};
    for (int64_t a : allowed)
        if (a == v) { *out = v; return true; }
    return false;
}
bool parse_bool(const char *s, int64_t *out) { return parse_one_of<0, 1>(s, out); }
template <int64_t LO, int64_t HI>
bool parse_range(const char *s, int64_t *out) { return parse_int(s, out) && *out >= LO && *out <= HI; }
bool parse_tile(const char *s, int64_t *out) { return parse_int(s, out) && *out >= 64 && *out % 64 == 0; }
// strictly "rays" | "samples" | "tiles" (anything else used to mean "samples")
bool parse_emit(const char *s, int64_t *out) {
    if (!strcmp(s, "rays")) { *out = 1; return true; }
    if (!strcmp(s, "samples")) { *out = 2; return true; }
    if (!strcmp(s, "tiles")) { *out = 3; return true; }
    return false;
}

const OptionSpec kSpecs[OPT_COUNT] = {
    {"e", "samples per lane of the tiled streaming kernels: 1 | 2 | 4", parse_one_of<1, 2, 4>},
    {"tile", "nominal tile of the tiled streaming kernels (a multiple of 64 * e)", parse_tile},
    {"split_p", "lanes per ray of the one-level count pass: 1 | 2 | 4 | 8 | 16", parse_one_of<1, 2, 4, 8, 16>},
    {"seg_p", "several levels, cone_angle = 0: lanes per ray of the segment count pass: 8 | 32", parse_one_of<8, 32>},
    {"cone_p", "lanes per ray of the cone-angle count pass: 8 | 16 | 32 | 64", parse_one_of<8, 16, 32, 64>},
    {"cone", "0: cone_angle != 0 takes the general lane-per-ray kernel", parse_bool},
    {"split_l2", "16 lanes per ray: grid image in LDS (0) / read from L2 (1)", parse_bool},
    {"count_l2", "lane-per-ray count and fill kernels: grid image in LDS (0) / from L2 (1)", parse_bool},
    {"emit", "emit pass: rays (16 lanes per ray walk its run records) | samples (a lane per sample) | tiles (a wave expands a block of rays)", parse_emit},
    {"scan_rw", "packed scan: rows per wave 4 | 16", parse_one_of<4, 16>},
    {"split_blk", "workgroup size of the 16-lane count pass: 256 | 512", parse_one_of<256, 512>},
    {"split_xt", "0: no crossing-time arrays in the 512-thread count pass", parse_bool},
    {"segments", "0: several levels take the lane-per-ray count pass", parse_bool},
    {"split_cap", "grids read from L2: entries of a part's boundary list, 16 | 24 | 32", parse_one_of<16, 24, 32>},
    {"emit_rb", "tile form of the emit pass: log2 of the rays per wave, 0 ... 6", parse_one_of<0, 1, 2, 3, 4, 5, 6>},
    {"chunk_prefetch", "0: sample_occgrid of the torch extension never launches the next ray slice's count pass ahead of its call", parse_bool},
    {"speculative_emit", "0: sample_occgrid of the torch extension launches the emit pass after the read-back", parse_bool},
    {"skip", "lane-per-ray lattice count pass: 0 = voxel by voxel, 1 = empty-space macro steps (brick distances from L2) (unset: a wave takes the macro steps when its rays are coherent)", parse_bool},
    {"fused_sample", "the sampling call as ONE launch (count, look-back over the workgroups and emit in the count kernel; 3072 ... 8192 rays of a one-level grid that fits LDS): 0 never, 1 when the caller's guess is at most 80 samples per ray (unset), 2 whatever the guess", parse_one_of<0, 1, 2>},
    {"fused_vis", "1: calls whose workgroups are all resident run the visibility filter as ONE launch (one pass, survivors staged in LDS and flushed behind a look-back over the workgroups); unset / 0: mask pass + compaction kernels (the single launch measures 3.5 us slower at the training size)", parse_bool},
    {"fold_fill", "0: nfa_rendering_fwd fills the rays without a sample with a launch of its own (unset: extra workgroups of its kernel do it for up to 2^20 samples, from the gaps between ascending ray_indices)", parse_bool},
    {"sync_spin_us", "bound of a look-back's wait inside the single-launch forms (nfa_traverse_sample, nfa_visibility_compact_sync, nfa_grid_occupied_cells), microseconds: 0 ... 10000000 (unset: 2000; 0: a look-back gives up at the first state that is not there yet, and the caller's fallback runs — tests)", parse_range<0, 10000000>},
};

int find_option(const char *name) {
    if (!name) return -1;
    std::string s(name);
    for (char &c : s) c = (char)tolower((unsigned char)c);
    if (s.rfind("nfa_", 0) == 0) s = s.substr(4);
    for (int i = 0; i < OPT_COUNT; ++i)
        if (s == kSpecs[i].name) return i;
    return -1;
}

struct Seeded {
    OptionTable live;
    int64_t at_load[OPT_COUNT];
    Seeded() {
        for (int i = 0; i < OPT_COUNT; ++i) {
            int64_t v = kOptUnset;
            std::string env = "NFA_";
            for (const char *c = kSpecs[i].name; *c; ++c) env += (char)toupper((unsigned char)*c);
            if (const char *e = getenv(env.c_str())) {
                if (!kSpecs[i].parse(e, &v)) v = kOptUnset;       // an unparsable variable means "auto", as before
            }
            // the one variable of round 3 that was spelled the other way round
            if (i == OPT_SPECULATIVE_EMIT && getenv("NFA_NO_SPECULATIVE_EMIT")) v = 0;
            at_load[i] = v;
            live.v[i].store(v, std::memory_order_relaxed);
        }
    }
};
Seeded &seeded() {
    static Seeded s;          // thread-safe one-time construction: the ONLY getenv calls of the library
    return s;
}

}  // namespace

OptionTable &option_table() { return seeded().live; }

}  // namespace nfa

using namespace nfa;

NFA_EXPORT int nfa_set_option(const char *name, const char *value) {
    const int i = find_option(name);
    NFA_REQUIRE(i >= 0, "nfa_set_option: unknown option '%s'", name ? name : "(null)");
    int64_t v = kOptUnset;
    if (value && value[0] && strcmp(value, "auto") != 0)
        NFA_REQUIRE(kSpecs[i].parse(value, &v), "nfa_set_option: '%s' is not a value of %s (%s)", value, kSpecs[i].name, kSpecs[i].doc);
    option_table().v[i].store(v, std::memory_order_relaxed);
    return NFA_OK;
}

NFA_EXPORT int nfa_get_option(const char *name, int64_t *value, int32_t *is_set) {
    const int i = find_option(name);
    NFA_REQUIRE(i >= 0, "nfa_get_option: unknown option '%s'", name ? name : "(null)");
    const int64_t v = option_table().v[i].load(std::memory_order_relaxed);
    if (is_set) *is_set = v != kOptUnset;
    if (value) *value = v == kOptUnset ? 0 : v;
    return NFA_OK;
}

NFA_EXPORT void nfa_reset_options(void) {
    Seeded &s = seeded();
    for (int i = 0; i < OPT_COUNT; ++i) s.live.v[i].store(s.at_load[i], std::memory_order_relaxed);
}

NFA_EXPORT int32_t nfa_option_count(void) { return OPT_COUNT; }
NFA_EXPORT const char *nfa_option_name(int32_t index) { return index >= 0 && index < OPT_COUNT ? kSpecs[index].name : nullptr; }
NFA_EXPORT const char *nfa_option_doc(int32_t index) { return index >= 0 && index < OPT_COUNT ? kSpecs[index].doc : nullptr; }
