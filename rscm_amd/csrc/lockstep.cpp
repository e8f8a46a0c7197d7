// The lock-step scheduler of component graphs: Model::step for a list of linked ensembles
// (crates/rscm-core/src/model/runtime.rs:504-527 walks the graph once per step; here the walk is cut into
// launches -- fused groups of light components (one launch for many steps when the whole graph is light), the
// heavy components' own kernels).
#include <algorithm>
#include <cstdlib>

#include "ens.hpp"
#include "experiment_env.hpp"

extern "C" {

// A/B switches and launch counters (include/rscm_gpu_internal.h).  Per calling thread: the boundary's model is
// one handle per device per thread, so two threads stepping two graphs neither race on the counters nor see
// each other's switches.
namespace {
struct LockstepSettings {
    bool fuse = true;        // consecutive light components of a step in one launch
    bool cache = true;       // multi-step fused launches keep per-member values in LDS between steps
    bool by_value = true;    // short op lists travel in the kernel arguments
    bool split = true;       // independent ops of a one-step segment on two wavefronts (group_split_kernel)
    bool merge = true;       // a step's last fused segment and the next step's first one in ONE launch (they are consecutive launches anyway)
    bool own_cut = true;     // the cut launch through the kernel compiled for its sequence of kinds where one exists (csrc/group.hip)
    int64_t merged_launches = 0;                // launches that carried two steps' segments since the last rscm_gpu_lockstep_merged_launches
    int64_t own_cut_launches = 0;               // cut launches that went through a kernel of their own since the last rscm_gpu_lockstep_own_cut_launches
    int32_t last_layout[3 + 2 * rscm::kGroupTableOps] = {};   // the last one-step by-value launch: n_ops, n_first, n_second, then kind and step offset per op
    int64_t launches = 0, component_steps = 0;  // since the thread's last rscm_gpu_lockstep_stats
    int64_t split_launches = 0;                 // of those, launches of group_split_kernel
};
thread_local LockstepSettings t_ls;
}  // namespace

int rscm_gpu_lockstep_stats(int64_t* launches, int64_t* component_steps)
{
    if (launches) *launches = t_ls.launches;
    if (component_steps) *component_steps = t_ls.component_steps;
    t_ls.launches = t_ls.component_steps = 0;
    return RSCM_OK;
}

int rscm_gpu_lockstep_split_launches(int64_t* out)
{
    if (out) *out = t_ls.split_launches;
    t_ls.split_launches = 0;
    return RSCM_OK;
}

int rscm_gpu_set_udeb_variant(int32_t variant)
{
    if (variant != -1 && variant != 0 && variant != 2 && variant != 3) return fail(RSCM_ERR_INVALID, "ClimateUDEB kernel variant must be -1, 0, 2 or 3");
    rscm::set_udeb_variant(variant);
    return RSCM_OK;
}

int rscm_gpu_set_run_plan(int32_t mode)
{
    if (mode < -1 || mode > 1) return fail(RSCM_ERR_INVALID, "run plan %d (-1 default, 0 one plain launch, 1 the two-stream cut)", mode);
    set_run_plan(mode);
    return RSCM_OK;
}

int rscm_gpu_derive_launches(int64_t* out)
{
    const int64_t n = take_derive_launches();
    if (out) *out = n;
    return RSCM_OK;
}

int rscm_gpu_experiments_build(void)
{
#ifdef RSCM_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

int rscm_gpu_fail_chunk_launch(int32_t k)
{
    if (k < 0) return fail(RSCM_ERR_INVALID, "chunk launch number %d (1-based; 0 turns the hook off)", k);
    set_fail_chunk_launch(k);
    return RSCM_OK;
}

int rscm_gpu_set_lockstep_fusion(int32_t enabled)
{
    if (enabled < 0 || enabled > 6) return fail(RSCM_ERR_INVALID, "lock-step fusion mode %d (0..6)", enabled);
    t_ls.fuse = enabled != 0;
    t_ls.cache = enabled == 1 || enabled >= 3;
    t_ls.by_value = enabled != 3;
    t_ls.split = enabled != 4;
    t_ls.merge = enabled != 5;   // (5: round 5's launch plan)
    t_ls.own_cut = enabled != 5 && enabled != 6;   // (6: merged launches through the op interpreter)
    return RSCM_OK;
}

int rscm_gpu_lockstep_own_cut_launches(int64_t* out)
{
    if (out) *out = t_ls.own_cut_launches;
    t_ls.own_cut_launches = 0;
    return RSCM_OK;
}

int rscm_gpu_lockstep_last_layout(int32_t* out)
{
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    memcpy(out, t_ls.last_layout, sizeof t_ls.last_layout);
    return RSCM_OK;
}

int rscm_gpu_lockstep_merged_launches(int64_t* out)
{
    if (out) *out = t_ls.merged_launches;
    t_ls.merged_launches = 0;
    return RSCM_OK;
}

// Kinds whose one-step launch the group kernel can absorb (csrc/group.hip): the light per-member
// components.  ClimateUDEB, OceanCarbon, HalocarbonChemistry and the fused coupled chain keep their own
// launches; GhgForcing joins only with linked concentrations (its table path uses host-built rows).
static bool fusable(const rscm_ens* h)
{
    switch (h->kind) {
        case RSCM_KIND_TWO_LAYER: case RSCM_KIND_OZONE_FORCING: case RSCM_KIND_AEROSOL_DIRECT: case RSCM_KIND_AEROSOL_INDIRECT:
        case RSCM_KIND_CH4_CHEMISTRY: case RSCM_KIND_N2O_CHEMISTRY: case RSCM_KIND_CO2_BUDGET: case RSCM_KIND_TERRESTRIAL_CARBON:
        case RSCM_KIND_FOURBOX_OHU: case RSCM_KIND_OSPP: case RSCM_KIND_CARBON_CYCLE: case RSCM_KIND_CO2_ERF: case RSCM_KIND_AGGREGATE:
            return true;
        case RSCM_KIND_GHG_FORCING: return h->n_linked > 0;
        default: return false;
    }
}

// the step range is an argument of the fused launch, not part of the table
static void clear_step_fields(rscm::GroupOp& op)
{
    switch (op.kind) {
        case RSCM_KIND_TWO_LAYER: op.u.tl.step_begin = op.u.tl.step_end = 0; break;
        case RSCM_KIND_GHG_FORCING: op.u.ghg.step_begin = op.u.ghg.step_end = 0; break;
        case RSCM_KIND_CH4_CHEMISTRY: case RSCM_KIND_N2O_CHEMISTRY: op.u.chem.step_begin = op.u.chem.step_end = 0; break;
        case RSCM_KIND_CO2_BUDGET: case RSCM_KIND_TERRESTRIAL_CARBON: case RSCM_KIND_CARBON_CYCLE:
            op.u.carbon.step_begin = op.u.carbon.step_end = 0; break;
        default: op.u.pw.step_begin = op.u.pw.step_end = 0; break;
    }
}

// LDS slots for a multi-step launch of a graph of light components (csrc/group.hip, CACHED): every op that
// can keep values there gets one slot per series (the latest row: its own state for the next step, and what
// its consumers read) and, while the budget lasts, one per parameter row if any of its rows varies over the
// members.  A link is served from the producer's slot when the value it wants is the one the slot holds at
// that point of the step: a producer earlier in the order read at n+1 (this step's value), or a producer
// later in the order read at n (what it left in the previous step -- not at a launch's first step).
static constexpr int32_t kCacheSlotBudget = 20;  // x 2 KiB per workgroup: four workgroups (16 wavefronts) per CU
static bool keeps_slots(int32_t kind)
{
    switch (kind) {
        case RSCM_KIND_TWO_LAYER: case RSCM_KIND_CARBON_CYCLE: case RSCM_KIND_AEROSOL_INDIRECT: case RSCM_KIND_FOURBOX_OHU:
        case RSCM_KIND_OSPP: case RSCM_KIND_CO2_ERF: case RSCM_KIND_AGGREGATE: case RSCM_KIND_CO2_BUDGET:
            return true;  // every kind the light variant of the group kernel runs
        default: return false;
    }
}
static int32_t assign_cache_slots(LockstepPlan* plan, int32_t first, int32_t count, std::vector<rscm::OpCache>& out, bool param_slots)
{
    out.assign((size_t)count, rscm::OpCache{});
    int32_t next = 0;
    for (int32_t k = 0; k < count; ++k) {
        rscm::OpCache& c = out[(size_t)k];
        c.series_slot = c.param_slot = -1;
        for (int32_t& s : c.link_slot) s = -1;
        c.link_warm = 0;
        const rscm_ens* h = plan->handles[first + k];
        const int32_t n_series = h->V - 1;
        if (keeps_slots(h->kind) && n_series > 0 && next + n_series <= kCacheSlotBudget) {
            c.series_slot = next;
            next += n_series;
        }
    }
    for (int32_t k = 0; k < count; ++k) {
        const rscm_ens* h = plan->handles[first + k];
        const uint64_t all_rows = h->P >= 64 ? ~0ull : ((1ull << h->P) - 1ull);
        const bool varies = (h->uniform_rows & all_rows) != all_rows;
        if (param_slots && keeps_slots(h->kind) && h->kind != RSCM_KIND_AGGREGATE && varies && h->P <= 16 && next + h->P <= kCacheSlotBudget) {
            out[(size_t)k].param_slot = next;
            next += h->P;
        }
    }
    for (int32_t k = 0; k < count; ++k) {
        const rscm_ens* h = plan->handles[first + k];
        if (!keeps_slots(h->kind)) continue;
        for (int32_t j = 0; j < rscm::kMaxLinks && j < h->n_inputs; ++j) {
            const auto& l = h->links[j];
            if (!l.src) continue;
            int32_t at = -1;
            for (int32_t q = 0; q < count; ++q)
                if (plan->handles[first + q] == l.src) at = q;
            if (at < 0 || out[(size_t)at].series_slot < 0 || l.var < 1 || l.var > l.src->V - 1) continue;
            const bool reads_end = h->kind == RSCM_KIND_AGGREGATE || l.off == 1;
            if (at < k ? !reads_end : reads_end) continue;  // the slot holds the other row at that point
            out[(size_t)k].link_slot[j] = out[(size_t)at].series_slot + (l.var - 1);
            if (at >= k) out[(size_t)k].link_warm |= 1u << j;
        }
    }
    return next;
}

// Two independent sets of ops + a tail for a one-step segment (csrc/group.hip, group_split_kernel).  Ops are tied together -- must run on
// the same wavefront, in their order -- whenever one reads the row the other writes in this step (a link read at n + 1, whichever of the
// two comes first in the order: a consumer that runs BEFORE its producer reads what the row held before, and must keep doing so); links
// read at n touch another row than the one being written and tie nothing.  For every tail start t the ops before it fall into connected
// sets; two bins are filled greedily by estimated cost; the cut with the shortest critical path wins if it beats the serial chain.
struct SplitPlan {
    int32_t order[rscm::kGroupTableOps];
    int32_t n_first = 0, n_second = 0;
};
static int32_t op_cost(int32_t kind)
{
    // (one unit ~ a dependent round trip to memory plus a few dozen instructions; the chemistry's Prather passes, the forcing
    // formulas' logarithms and the RK4 box models weigh by their instruction counts on top)
    switch (kind) {
        case RSCM_KIND_TWO_LAYER: return 12;
        case RSCM_KIND_CH4_CHEMISTRY: return 6;
        case RSCM_KIND_N2O_CHEMISTRY: return 5;
        case RSCM_KIND_CARBON_CYCLE: case RSCM_KIND_GHG_FORCING: case RSCM_KIND_TERRESTRIAL_CARBON: return 4;
        case RSCM_KIND_OZONE_FORCING: case RSCM_KIND_AEROSOL_DIRECT: return 2;
        default: return 1;
    }
}
// idx[k]: the handle of op k in the plan; off[k]: 0 = the launch's step n, 1 = step n + 1 (a merged launch).  Op k reads row
// s_k + (1 if it reads at the end of its step, else l.off) of a producer q, which writes row s_q + 1: tied when the two are the same row.
static bool plan_split(const LockstepPlan* plan, const int32_t* idx, const int32_t* off, int32_t count, SplitPlan* out)
{
    static const bool enabled = rscm::experiment_env("RSCM_LOCKSTEP_SPLIT", 1) != 0;   // (experiments build only: 0 = A/B runs without it)
    if (!enabled || !t_ls.split || count < 3 || count > rscm::kGroupTableOps) return false;
    bool tie[rscm::kGroupTableOps][rscm::kGroupTableOps] = {};
    int32_t cost[rscm::kGroupTableOps], serial = 0;
    for (int32_t k = 0; k < count; ++k) {
        const rscm_ens* h = plan->handles[idx[k]];
        cost[k] = op_cost(h->kind);
        serial += cost[k];
        for (int32_t j = 0; j < rscm::kMaxLinks && j < h->n_inputs; ++j) {
            const auto& l = h->links[j];
            if (!l.src) continue;
            const bool reads_end = h->kind == RSCM_KIND_AGGREGATE || l.off == 1;
            const int32_t reads_row = off[k] + (reads_end ? 1 : 0);
            for (int32_t q = 0; q < count; ++q)
                if (plan->handles[idx[q]] == l.src && q != k && reads_row == off[q] + 1) tie[k][q] = tie[q][k] = true;
        }
    }
    // (worth it from ~15 % of the serial chain: the cut doubles the wavefronts that must be resident)
    int32_t best = serial - std::max(1, (serial * 3 + 19) / 20), best_t = -1, best_bin[rscm::kGroupTableOps] = {};
    for (int32_t t = 2; t <= count; ++t) {
        int32_t comp[rscm::kGroupTableOps];
        for (int32_t k = 0; k < t; ++k) comp[k] = k;
        for (bool changed = true; changed;) {   // connected sets of the ops before the tail (at most eight ops)
            changed = false;
            for (int32_t a = 0; a < t; ++a)
                for (int32_t b = 0; b < t; ++b)
                    if (tie[a][b] && comp[a] != comp[b]) {
                        const int32_t lo = std::min(comp[a], comp[b]), hi = std::max(comp[a], comp[b]);
                        for (int32_t k = 0; k < t; ++k)
                            if (comp[k] == hi) comp[k] = lo;
                        changed = true;
                    }
        }
        int32_t weight[rscm::kGroupTableOps] = {}, bin_of[rscm::kGroupTableOps], load[2] = {0, 0}, n_sets = 0;
        for (int32_t k = 0; k < t; ++k) weight[comp[k]] += cost[k];
        for (int32_t c = 0; c < t; ++c) {
            bin_of[c] = -1;
            if (weight[c] > 0) ++n_sets;
        }
        if (n_sets < 2) continue;
        for (int32_t placed = 0; placed < n_sets; ++placed) {   // heaviest set first, into the lighter bin
            int32_t pick = -1;
            for (int32_t c = 0; c < t; ++c)
                if (weight[c] > 0 && bin_of[c] < 0 && (pick < 0 || weight[c] > weight[pick])) pick = c;
            const int32_t b = load[0] <= load[1] ? 0 : 1;
            bin_of[pick] = b;
            load[b] += weight[pick];
        }
        int32_t tail = 0;
        for (int32_t k = t; k < count; ++k) tail += cost[k];
        const int32_t path = std::max(load[0], load[1]) + tail + 1;   // + the barrier
        if (path <= best && (best_t < 0 || path < best)) {
            best = path;
            best_t = t;
            for (int32_t k = 0; k < t; ++k) best_bin[k] = bin_of[comp[k]];
        }
    }
    if (best_t < 0) return false;
    int32_t pos = 0;
    out->n_first = out->n_second = 0;
    for (int32_t b = 0; b < 2; ++b)
        for (int32_t k = 0; k < best_t; ++k)
            if (best_bin[k] == b) {
                out->order[pos++] = k;
                ++(b == 0 ? out->n_first : out->n_second);
            }
    for (int32_t k = best_t; k < count; ++k) out->order[pos++] = k;
    return out->n_first > 0 && out->n_second > 0;
}

// Model steps [n, n + len) of handles [first, first + count) of the plan as ONE launch.  len > 1 only when
// the segment is the whole graph: then nothing outside the launch reads or writes between its steps.
// next_count > 0 (len == 1, by-value table only): a MERGED launch -- the handles [next_first, next_first + next_count) ride along at
// step n + 1.  They are the first segment of the next step, i.e. the launch that would follow this one anyway: same order, same
// operands, one launch boundary less, and the two segments' independent ops share the two wavefronts of the split kernel.
static int fused_segment(LockstepPlan* plan, int32_t first, int32_t count, int32_t n, int32_t len, int32_t next_first = 0, int32_t next_count = 0)
{
    rscm_ens* lead = plan->handles[first];
    const int32_t total = count + next_count;
    if (next_count > 0 && (len != 1 || total > rscm::kGroupTableOps || !t_ls.by_value)) return fail(RSCM_ERR_STATE, "a merged launch needs a by-value table");
    int32_t idx[rscm::kMaxGroupOps + rscm::kGroupTableOps], off[rscm::kMaxGroupOps + rscm::kGroupTableOps];
    for (int32_t k = 0; k < total; ++k) {
        idx[k] = k < count ? first + k : next_first + (k - count);
        off[k] = k < count ? 0 : 1;
    }
    bool all_small = true;
    for (int32_t k = 0; k < total; ++k) all_small = all_small && rscm::group_kind_is_small(plan->handles[idx[k]]->kind);
    std::vector<rscm::OpCache> slots;
    int32_t cache_slots = 0;
    // a multi-step launch of a graph whose sequence of kinds has a kernel of its own (csrc/group.hip, group_seq_kernel): that kernel keeps
    // the parameters in registers, so only the series get LDS slots
    bool own_kernel = false;
    if (len > 1 && all_small && t_ls.cache && t_ls.by_value && count <= rscm::kGroupTableOps) {
        int32_t kinds[rscm::kGroupTableOps];
        for (int32_t k = 0; k < count; ++k) kinds[k] = plan->handles[first + k]->kind;
        own_kernel = rscm::group_seq_available(kinds, count);
    }
    if (len > 1 && all_small && t_ls.cache) cache_slots = assign_cache_slots(plan, first, count, slots, !own_kernel);
    if (cache_slots <= 0) own_kernel = false;
    for (int32_t k = 0; k < total; ++k) {
        rscm_ens* h = plan->handles[idx[k]];
        if (int rc = step_check(h, n + off[k], n + off[k] + len)) return rc;
        if (int rc = step_window_pre(h, n + off[k], n + off[k] + len)) return rc;
    }
    // a short op list travels by value in the kernel arguments (one-step launches: window slides change pointers
    // every few steps); a longer one, and the multi-step launch with LDS slots, through the device table, of which
    // only what changed since the last launch is uploaded
    const bool by_value = (t_ls.by_value && total <= rscm::kGroupTableOps && cache_slots == 0) || own_kernel;
    rscm::GroupTable table;
    if (by_value) memset((void*)&table, 0, sizeof table);
    for (int32_t k = 0; k < total; ++k) {
        rscm_ens* h = plan->handles[idx[k]];
        const int32_t at = n + off[k];
        rscm::InputLinks links{};
        int32_t linked = 0;
        if (int rc = step_links(h, at, at + 1, links, linked)) return rc;
        rscm::GroupOp op;
        memset((void*)&op, 0, sizeof op);
        if (int rc = step_launch(h, at, at + 1, links, linked, &op)) return rc;
        if (op.kind < 0) return fail(RSCM_ERR_STATE, "handle %d (kind %d) cannot be fused", idx[k], h->kind);
        clear_step_fields(op);
        op.step_off = off[k];
        if (cache_slots > 0) {
            op.cache = slots[(size_t)k];
        } else {
            op.cache.series_slot = op.cache.param_slot = -1;
            for (int32_t& sl : op.cache.link_slot) sl = -1;
        }
        if (by_value) {
            memcpy((void*)&table.ops[k], &op, sizeof op);
        } else if (!plan->valid[idx[k]] || memcmp(&plan->cached[idx[k]], &op, sizeof op) != 0) {
            if (plan->ring_pos == LockstepPlan::kRing) {  // every slot may still be the source of a queued copy
                HIPCHK(hipStreamSynchronize(lead->stream));
                plan->ring_pos = 0;
            }
            rscm::GroupOp* slot = plan->staging + plan->ring_pos++;
            memcpy((void*)slot, &op, sizeof op);
            HIPCHK(hipMemcpyAsync(plan->d_ops + idx[k], slot, sizeof op, hipMemcpyHostToDevice, lead->stream));
            memcpy((void*)&plan->cached[idx[k]], &op, sizeof op);
            plan->valid[idx[k]] = 1;
        }
        h->time_index = at + 1;  // provisional: later handles of the launch may read this one's row at + 1
    }
    hipError_t seq_status = hipSuccess;
    if (own_kernel) {
        if (!rscm::launch_group_seq(table, count, lead->N, n, n + len, cache_slots, lead->stream, &seq_status))
            return fail(RSCM_ERR_STATE, "no kernel for this sequence of kinds after all");
        HIPCHK(seq_status);
    } else {
        SplitPlan split;
        if (by_value && len == 1 && cache_slots == 0 && plan_split(plan, idx, off, total, &split)) {
            rscm::GroupTable ordered;
            memset((void*)&ordered, 0, sizeof ordered);
            for (int32_t k = 0; k < total; ++k) memcpy((void*)&ordered.ops[k], &table.ops[split.order[k]], sizeof(rscm::GroupOp));
            t_ls.last_layout[0] = total; t_ls.last_layout[1] = split.n_first; t_ls.last_layout[2] = split.n_second;
            for (int32_t k = 0; k < total; ++k) {
                t_ls.last_layout[3 + 2 * k] = ordered.ops[k].kind;
                t_ls.last_layout[4 + 2 * k] = ordered.ops[k].step_off | (ordered.ops[k].variant << 8);
            }
            HIPCHK(rscm::launch_group_split(ordered, split.n_first, split.n_second, total, lead->N, n, all_small, lead->stream, t_ls.own_cut));
            if (t_ls.own_cut && rscm::group_split_seq_available(ordered, split.n_first, split.n_second, total)) t_ls.own_cut_launches += 1;
            t_ls.split_launches += 1;
        } else {
            HIPCHK(rscm::launch_group(by_value ? nullptr : plan->d_ops + first, by_value ? &table : nullptr, total, lead->N, n, n + len, all_small,
                                      cache_slots, lead->stream));
        }
    }
    if (next_count > 0) t_ls.merged_launches += 1;
    for (int32_t k = 0; k < total; ++k)
        if (int rc = step_finish(plan->handles[idx[k]], n + off[k], n + off[k] + len)) return rc;
    return RSCM_OK;
}

int rscm_ens_run_lockstep(rscm_ens* const* handles, int32_t n_handles, int32_t step_begin, int32_t step_end)
{
    GUARD_BEGIN
    if (!handles || n_handles < 1) return fail(RSCM_ERR_INVALID, "need at least one handle");
    for (int32_t k = 0; k < n_handles; ++k) {
        if (!handles[k]) return fail(RSCM_ERR_INVALID, "handle %d is NULL", k);
        if (handles[k]->stream != handles[0]->stream)
            return fail(RSCM_ERR_STATE, "handle %d runs on another stream than handle 0", k);
        if (handles[k]->time_index != step_begin)
            return fail(RSCM_ERR_STATE, "handle %d is at time index %d, not at step_begin %d", k, handles[k]->time_index, step_begin);
    }
    // Consecutive fusable components become one launch per step (csrc/group.hip); the others, and
    // fusable ones on their own, keep their kernels.
    std::vector<std::pair<int32_t, int32_t>> segments;  // (first, count)
    for (int32_t k = 0; k < n_handles;) {
        int32_t c = 1;
        if (t_ls.fuse && fusable(handles[k]))
            while (k + c < n_handles && c < rscm::kMaxGroupOps && fusable(handles[k + c]) && handles[k + c]->N == handles[k]->N &&
                   handles[k + c]->device == handles[k]->device)
                ++c;
        segments.emplace_back(k, c);
        k += c;
    }
    bool any_fused = false;
    for (const auto& sgm : segments) any_fused = any_fused || sgm.second > 1;
    LockstepPlan* plan = nullptr;
    if (any_fused) {
        rscm_ens* lead = handles[0];
        if (int rc = set_device(lead)) return rc;
        plan = lead->plan;
        const std::vector<rscm_ens*> list(handles, handles + n_handles);
        if (!plan || plan->handles != list) {
            if (!plan) plan = lead->plan = new LockstepPlan();
            HIPCHK(hipStreamSynchronize(lead->stream));
            HIPCHK(hipFree(plan->d_ops));
            plan->d_ops = nullptr;
            plan->handles = list;
            plan->cached.assign((size_t)n_handles, rscm::GroupOp());
            plan->valid.assign((size_t)n_handles, 0);
            plan->ring_pos = 0;
            HIPCHK(rscm::dev_malloc(&plan->d_ops, (size_t)n_handles * sizeof(rscm::GroupOp)));
            if (!plan->staging) HIPCHK(hipHostMalloc((void**)&plan->staging, LockstepPlan::kRing * sizeof(rscm::GroupOp), hipHostMallocDefault));
        }
    }
    // The window upkeep of the graph's handles (slides at the end of the step that fills a window, output rows) is collected while
    // a step -- or a chunk of steps in one launch -- is enqueued and issued in one or two launches at its end.
    WindowDeferral deferral;
    struct DeferGuard {
        rscm_ens* const* hs; int32_t n;
        DeferGuard(rscm_ens* const* h, int32_t k, WindowDeferral* d) : hs(h), n(k) { for (int32_t i = 0; i < n; ++i) hs[i]->defer = d; }
        ~DeferGuard() { for (int32_t i = 0; i < n; ++i) { hs[i]->defer = nullptr; hs[i]->derived_hold = false; } }
    } guard(handles, n_handles, t_ls.fuse ? &deferral : nullptr);
    // The member constants of every handle once per call, also where the caller holds the parameter block's device pointer (such a
    // block is re-derived before every run: here "run" is this call, not each of its one-step launches).
    for (int32_t k = 0; k < n_handles; ++k) {
        if (handles[k]->params_set)
            if (int rc = ensure_derived(handles[k])) return rc;
        handles[k]->derived_hold = true;
    }
    if (segments.size() == 1 && segments[0].second > 1) {
        // The whole graph is one fused segment: many model steps per launch.  A chunk ends where a windowed
        // handle runs out of rows (its window slides between launches).
        for (int32_t n = step_begin; n < step_end;) {
            int32_t len = step_end - n;
            for (int32_t k = 0; k < n_handles; ++k) {
                const rscm_ens* h = handles[k];
                if (h->windowed) len = std::min(len, std::max(1, h->rows - h->keep_rows()));
            }
            t_ls.launches += 1;
            t_ls.component_steps += (int64_t)n_handles * len;
            if (int rc = fused_segment(plan, 0, n_handles, n, len)) return rc;
            if (int rc = window_flush(&deferral, handles[0]->stream)) return rc;
            n += len;
        }
        return RSCM_OK;
    }
    auto run_one = [&](const std::pair<int32_t, int32_t>& sgm, int32_t n) -> int {
        t_ls.launches += 1;
        t_ls.component_steps += sgm.second;
        if (sgm.second > 1) return fused_segment(plan, sgm.first, sgm.second, n, 1);
        return run_range(handles[sgm.first], n, n + 1, false);
    };
    // MERGED schedule.  In graph order a step's last segment L(n) and the next step's first segment F(n + 1) are consecutive launches
    // with nothing but the window upkeep between them.  Where both are fused (light) segments and fit one by-value table, they go
    // out as ONE launch: the same ops on the same operands in the same order, one launch boundary less per step, and their
    // independent ops side by side on the two wavefronts of the split kernel.  What moves is the upkeep: it is flushed after the
    // merged launch (L's handles have finished step n, F's step n + 1) instead of between the two.  Every window keeps at least
    // two rows behind its handle's time index (keep_rows), so the rows n and n + 1 that the other components read of a handle that
    // is one step ahead are resident whichever side of the flush they are read on.  Not merged: when an op of F reads a row at the
    // END of its step from a handle outside F (a read-ahead link: the window of that handle would have to have moved first).
    bool merged = false;
    if (t_ls.fuse && t_ls.merge && t_ls.by_value && plan && segments.size() >= 3 && step_end - step_begin >= 2) {
        const auto& F = segments.front();
        const auto& L = segments.back();
        merged = fusable(handles[F.first]) && fusable(handles[L.first]) && F.second + L.second <= rscm::kGroupTableOps &&
                 handles[F.first]->N == handles[L.first]->N && handles[F.first]->device == handles[L.first]->device;
        for (int32_t k = F.first; merged && k < F.first + F.second; ++k) {
            const rscm_ens* h = handles[k];
            for (int32_t j = 0; j < rscm::kMaxLinks && j < h->n_inputs; ++j) {
                const auto& l = h->links[j];
                if (!l.src || !(h->kind == RSCM_KIND_AGGREGATE || l.off == 1)) continue;
                bool inside = false;
                for (int32_t q = F.first; q < F.first + F.second; ++q) inside = inside || handles[q] == l.src;
                if (!inside) merged = false;
            }
        }
    }
    if (merged) {
        const auto F = segments.front(), L = segments.back();
        if (int rc = run_one(F, step_begin)) return rc;                         // prologue: F(step_begin)
        if (int rc = window_flush(&deferral, handles[0]->stream)) return rc;
        for (int32_t n = step_begin; n < step_end; ++n) {
            for (size_t q = 1; q + 1 < segments.size(); ++q)
                if (int rc = run_one(segments[q], n)) return rc;
            if (n + 1 < step_end) {
                t_ls.launches += 1;
                t_ls.component_steps += L.second + F.second;
                if (int rc = fused_segment(plan, L.first, L.second, n, 1, F.first, F.second)) return rc;   // L(n) + F(n + 1)
            } else if (int rc = run_one(L, n)) {                                // epilogue: L(step_end - 1) alone
                return rc;
            }
            if (int rc = window_flush(&deferral, handles[0]->stream)) return rc;
        }
        return RSCM_OK;
    }
    for (int32_t n = step_begin; n < step_end; ++n) {
        for (const auto& sgm : segments)
            if (int rc = run_one(sgm, n)) return rc;
        if (int rc = window_flush(&deferral, handles[0]->stream)) return rc;
    }
    return RSCM_OK;
    GUARD_END
}

}  // extern "C"
