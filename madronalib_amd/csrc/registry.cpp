// registry.cpp — processors and ops by NAME. The reference's dynamic-proc stub registers a class under a
// string ("multiply": static ProcRegistryEntry<ProcMultiply> classReg("multiply"), source/procs/MLProcMultiply.cpp:44-47)
// and names its params, inputs and outputs with strings (paramNames / inputNames / outputNames, :12-18; process() reads
// input("foo"), writes output("baz"), :29-32). Here every node kind a graph can hold has such an entry - name, input names,
// parameter (coefficient) names, output name - so a host can describe a whole patch with strings (a preset file, a UI) and
// never touch an enum: mlgpu_graph_add_named adds a node by registry name with inputs given by node name,
// mlgpu_graph_set_named_coeff / mlgpu_graph_set_param_by_name fill per-voice tables by name. Host code only.
#include <string.h>

#include <string>

#include "mlgpu_internal.hpp"

namespace
{
struct Entry
{
  const char* name;
  int nodeType;  // MLGPU_REGISTRY_OP / _PROC / _VOP
  int kind;
  const char* inputs;  // comma-separated; a trailing '?' marks an optional input (the modulated forms)
  const char* params;  // comma-separated coefficient names, in coefficient-slot order
  const char* output;
};

// names: the reference's function / class name in lower_snake_case; inputs as its signature names them
const Entry kEntries[] = {
    // ---- stateless ops, MLDSPOps.h:570-917 -------------------------------------------------------------------
    {"sqrt", MLGPU_REGISTRY_OP, MLGPU_OP_SQRT, "x", "", "out"},
    {"sqrt_approx", MLGPU_REGISTRY_OP, MLGPU_OP_SQRT_APPROX, "x", "", "out"},
    {"abs", MLGPU_REGISTRY_OP, MLGPU_OP_ABS, "x", "", "out"},
    {"sign", MLGPU_REGISTRY_OP, MLGPU_OP_SIGN, "x", "", "out"},
    {"sign_bit", MLGPU_REGISTRY_OP, MLGPU_OP_SIGN_BIT, "x", "", "out"},
    {"sin", MLGPU_REGISTRY_OP, MLGPU_OP_SIN, "x", "", "out"},
    {"cos", MLGPU_REGISTRY_OP, MLGPU_OP_COS, "x", "", "out"},
    {"log", MLGPU_REGISTRY_OP, MLGPU_OP_LOG, "x", "", "out"},
    {"exp", MLGPU_REGISTRY_OP, MLGPU_OP_EXP, "x", "", "out"},
    {"log2", MLGPU_REGISTRY_OP, MLGPU_OP_LOG2, "x", "", "out"},
    {"exp2", MLGPU_REGISTRY_OP, MLGPU_OP_EXP2, "x", "", "out"},
    {"sin_approx", MLGPU_REGISTRY_OP, MLGPU_OP_SIN_APPROX, "x", "", "out"},
    {"cos_approx", MLGPU_REGISTRY_OP, MLGPU_OP_COS_APPROX, "x", "", "out"},
    {"exp_approx", MLGPU_REGISTRY_OP, MLGPU_OP_EXP_APPROX, "x", "", "out"},
    {"log_approx", MLGPU_REGISTRY_OP, MLGPU_OP_LOG_APPROX, "x", "", "out"},
    {"log2_approx", MLGPU_REGISTRY_OP, MLGPU_OP_LOG2_APPROX, "x", "", "out"},
    {"exp2_approx", MLGPU_REGISTRY_OP, MLGPU_OP_EXP2_APPROX, "x", "", "out"},
    {"fractional_part", MLGPU_REGISTRY_OP, MLGPU_OP_FRACTIONAL_PART, "x", "", "out"},
    {"round_float_to_int", MLGPU_REGISTRY_OP, MLGPU_OP_ROUND_FLOAT_TO_INT, "x", "", "out"},
    {"truncate_float_to_int", MLGPU_REGISTRY_OP, MLGPU_OP_TRUNCATE_FLOAT_TO_INT, "x", "", "out"},
    {"int_to_float", MLGPU_REGISTRY_OP, MLGPU_OP_INT_TO_FLOAT, "x", "", "out"},
    {"unsigned_int_to_float", MLGPU_REGISTRY_OP, MLGPU_OP_UNSIGNED_INT_TO_FLOAT, "x", "", "out"},
    {"phasor_to_sine", MLGPU_REGISTRY_OP, MLGPU_OP_PHASOR_TO_SINE, "phasor", "", "out"},
    {"add", MLGPU_REGISTRY_OP, MLGPU_OP_ADD, "in1,in2", "", "out"},
    {"subtract", MLGPU_REGISTRY_OP, MLGPU_OP_SUBTRACT, "in1,in2", "", "out"},
    {"multiply", MLGPU_REGISTRY_OP, MLGPU_OP_MULTIPLY, "in1,in2", "", "out"},
    {"divide", MLGPU_REGISTRY_OP, MLGPU_OP_DIVIDE, "in1,in2", "", "out"},
    {"divide_approx", MLGPU_REGISTRY_OP, MLGPU_OP_DIVIDE_APPROX, "in1,in2", "", "out"},
    {"pow", MLGPU_REGISTRY_OP, MLGPU_OP_POW, "base,exponent", "", "out"},
    {"pow_approx", MLGPU_REGISTRY_OP, MLGPU_OP_POW_APPROX, "base,exponent", "", "out"},
    {"min", MLGPU_REGISTRY_OP, MLGPU_OP_MIN, "in1,in2", "", "out"},
    {"max", MLGPU_REGISTRY_OP, MLGPU_OP_MAX, "in1,in2", "", "out"},
    {"add_int32", MLGPU_REGISTRY_OP, MLGPU_OP_ADD_INT32, "in1,in2", "", "out"},
    {"subtract_int32", MLGPU_REGISTRY_OP, MLGPU_OP_SUBTRACT_INT32, "in1,in2", "", "out"},
    {"equal", MLGPU_REGISTRY_OP, MLGPU_OP_EQUAL, "in1,in2", "", "mask"},
    {"not_equal", MLGPU_REGISTRY_OP, MLGPU_OP_NOT_EQUAL, "in1,in2", "", "mask"},
    {"greater_than", MLGPU_REGISTRY_OP, MLGPU_OP_GREATER_THAN, "in1,in2", "", "mask"},
    {"greater_than_or_equal", MLGPU_REGISTRY_OP, MLGPU_OP_GREATER_THAN_OR_EQUAL, "in1,in2", "", "mask"},
    {"less_than", MLGPU_REGISTRY_OP, MLGPU_OP_LESS_THAN, "in1,in2", "", "mask"},
    {"less_than_or_equal", MLGPU_REGISTRY_OP, MLGPU_OP_LESS_THAN_OR_EQUAL, "in1,in2", "", "mask"},
    {"phasor_to_saw", MLGPU_REGISTRY_OP, MLGPU_OP_PHASOR_TO_SAW, "phasor,freq", "", "out"},
    {"lerp", MLGPU_REGISTRY_OP, MLGPU_OP_LERP, "a,b,mix", "", "out"},
    {"inverse_lerp", MLGPU_REGISTRY_OP, MLGPU_OP_INVERSE_LERP, "a,b,x", "", "out"},
    {"clamp", MLGPU_REGISTRY_OP, MLGPU_OP_CLAMP, "x,min,max", "", "out"},
    {"within", MLGPU_REGISTRY_OP, MLGPU_OP_WITHIN, "x,min,max", "", "mask"},
    {"select", MLGPU_REGISTRY_OP, MLGPU_OP_SELECT, "a,b,mask", "", "out"},
    {"select_int", MLGPU_REGISTRY_OP, MLGPU_OP_SELECT_INT, "a,b,mask", "", "out"},
    {"phasor_to_pulse", MLGPU_REGISTRY_OP, MLGPU_OP_PHASOR_TO_PULSE, "phasor,freq,width", "", "out"},
    // ---- index generators, MLDSPOps.h:962-990 -----------------------------------------------------------------
    {"column_index", MLGPU_REGISTRY_VOP, MLGPU_VOP_COLUMN_INDEX, "", "", "out"},
    {"range_open", MLGPU_REGISTRY_VOP, MLGPU_VOP_RANGE_OPEN, "start,end", "", "out"},
    {"range_closed", MLGPU_REGISTRY_VOP, MLGPU_VOP_RANGE_CLOSED, "start,end", "", "out"},
    {"interpolate_dsp_vector_linear", MLGPU_REGISTRY_VOP, MLGPU_VOP_INTERPOLATE_LINEAR, "start,end", "", "out"},
    // ---- generators, MLDSPGens.h -------------------------------------------------------------------------------
    {"phasor_gen", MLGPU_REGISTRY_PROC, MLGPU_PROC_PHASOR_GEN, "freq", "", "out"},
    {"sine_gen", MLGPU_REGISTRY_PROC, MLGPU_PROC_SINE_GEN, "freq", "", "out"},
    {"saw_gen", MLGPU_REGISTRY_PROC, MLGPU_PROC_SAW_GEN, "freq", "", "out"},
    {"pulse_gen", MLGPU_REGISTRY_PROC, MLGPU_PROC_PULSE_GEN, "freq,width?", "width", "out"},
    {"noise_gen", MLGPU_REGISTRY_PROC, MLGPU_PROC_NOISE_GEN, "", "", "out"},
    {"tick_gen", MLGPU_REGISTRY_PROC, MLGPU_PROC_TICK_GEN, "freq", "", "out"},
    {"impulse_gen", MLGPU_REGISTRY_PROC, MLGPU_PROC_IMPULSE_GEN, "freq", "", "out"},
    {"one_shot_gen", MLGPU_REGISTRY_PROC, MLGPU_PROC_ONE_SHOT_GEN, "freq", "", "out"},
    {"test_sine_gen", MLGPU_REGISTRY_PROC, MLGPU_PROC_TEST_SINE_GEN, "freq", "", "out"},
    // ---- filters, MLDSPFilters.h -------------------------------------------------------------------------------
    {"lopass", MLGPU_REGISTRY_PROC, MLGPU_PROC_LOPASS, "in,omega?,k?", "g0,g1,g2", "out"},
    {"hipass", MLGPU_REGISTRY_PROC, MLGPU_PROC_HIPASS, "in", "g0,g1,g2,k", "out"},
    {"bandpass", MLGPU_REGISTRY_PROC, MLGPU_PROC_BANDPASS, "in", "g0,g1,g2", "out"},
    {"lo_shelf", MLGPU_REGISTRY_PROC, MLGPU_PROC_LO_SHELF, "in", "a1,a2,a3,m1,m2", "out"},
    {"hi_shelf", MLGPU_REGISTRY_PROC, MLGPU_PROC_HI_SHELF, "in", "a1,a2,a3,m0,m1,m2", "out"},
    {"bell", MLGPU_REGISTRY_PROC, MLGPU_PROC_BELL, "in", "a1,a2,a3,m1", "out"},
    {"one_pole", MLGPU_REGISTRY_PROC, MLGPU_PROC_ONE_POLE, "in", "a0,b1", "out"},
    {"dc_blocker", MLGPU_REGISTRY_PROC, MLGPU_PROC_DC_BLOCKER, "in", "coeff", "out"},
    {"differentiator", MLGPU_REGISTRY_PROC, MLGPU_PROC_DIFFERENTIATOR, "in", "", "out"},
    {"integrator", MLGPU_REGISTRY_PROC, MLGPU_PROC_INTEGRATOR, "in", "leak", "out"},
    {"peak", MLGPU_REGISTRY_PROC, MLGPU_PROC_PEAK, "in", "a0,b1,peak_hold_samples", "out"},
    {"rms", MLGPU_REGISTRY_PROC, MLGPU_PROC_RMS, "in", "a0,b1", "out"},
    {"adsr", MLGPU_REGISTRY_PROC, MLGPU_PROC_ADSR, "gate", "ka,kd,s,kr", "out"},
    {"gain", MLGPU_REGISTRY_PROC, MLGPU_PROC_GAIN, "in", "gain", "out"},
    // ---- control-rate to audio-rate, MLDSPGens.h:404-590 ------------------------------------------------------
    {"interpolator1", MLGPU_REGISTRY_PROC, MLGPU_PROC_INTERPOLATOR1, "target", "", "out"},
    {"linear_glide", MLGPU_REGISTRY_PROC, MLGPU_PROC_LINEAR_GLIDE, "target", "vectors_per_glide,dy_per_vector", "out"},
    {"sample_accurate_linear_glide", MLGPU_REGISTRY_PROC, MLGPU_PROC_SAMPLE_ACCURATE_LINEAR_GLIDE, "target", "samples_per_glide,dy_per_sample", "out"},
    // ---- delay lines, MLDSPFilters.h:799-1106 ------------------------------------------------------------------
    {"integer_delay", MLGPU_REGISTRY_PROC, MLGPU_PROC_INTEGER_DELAY, "in,delay?", "", "out"},
    {"allpass1", MLGPU_REGISTRY_PROC, MLGPU_PROC_ALLPASS1, "in", "coeff", "out"},
    {"fractional_delay", MLGPU_REGISTRY_PROC, MLGPU_PROC_FRACTIONAL_DELAY, "in,delay?", "", "out"},
    {"pitchbendable_delay", MLGPU_REGISTRY_PROC, MLGPU_PROC_PITCHBENDABLE_DELAY, "in,delay", "", "out"},
    {"tempo_lock", MLGPU_REGISTRY_PROC, MLGPU_PROC_TEMPO_LOCK, "phasor,dydx,isr", "", "out"},  // (x, float dydx, float isr): the floats are control-rate nodes
};
constexpr int kNumEntries = (int)(sizeof(kEntries) / sizeof(kEntries[0]));

int countList(const char* list)
{
  if (!list || !*list) return 0;
  int n = 1;
  for (const char* p = list; *p; ++p) n += (*p == ',');
  return n;
}
// the idx-th name of a comma-separated list (without the optional marker) into buf; false when there is none
bool itemOf(const char* list, int idx, char* buf, size_t bufLen, bool* optional)
{
  if (!list || !*list || idx < 0) return false;
  const char* p = list;
  for (int i = 0; i < idx; ++i)
  {
    p = strchr(p, ',');
    if (!p) return false;
    ++p;
  }
  const char* e = strchr(p, ',');
  size_t n = e ? (size_t)(e - p) : strlen(p);
  const bool opt = n && p[n - 1] == '?';
  if (opt) --n;
  if (optional) *optional = opt;
  if (buf && bufLen)
  {
    if (n >= bufLen) n = bufLen - 1;
    memcpy(buf, p, n);
    buf[n] = 0;
  }
  return true;
}
void fill(const Entry& e, mlgpu_registry_entry* out)
{
  out->name = e.name;
  out->node_type = e.nodeType;
  out->kind = e.kind;
  out->n_inputs = countList(e.inputs);
  out->n_required_inputs = 0;
  for (int i = 0; i < out->n_inputs; ++i)
  {
    bool opt = false;
    itemOf(e.inputs, i, nullptr, 0, &opt);
    if (!opt) ++out->n_required_inputs;
  }
  out->n_params = countList(e.params);
  out->output_name = e.output;
}
const Entry* find(const char* name)
{
  if (!name) return nullptr;
  for (const Entry& e : kEntries)
    if (!strcmp(e.name, name)) return &e;
  return nullptr;
}
}  // namespace

extern "C"
{
  int mlgpu_registry_count(void) { return kNumEntries; }

  int mlgpu_registry_get(int index, mlgpu_registry_entry* out)
  {
    if (!out || index < 0 || index >= kNumEntries) return MLGPU_ERR_RANGE;
    fill(kEntries[index], out);
    return MLGPU_OK;
  }

  int mlgpu_registry_lookup(const char* name, mlgpu_registry_entry* out)
  {
    const Entry* e = find(name);
    if (!e) return MLGPU_ERR_RANGE;
    if (out) fill(*e, out);
    return MLGPU_OK;
  }

  int mlgpu_registry_input_name(const char* name, int index, char* buf, size_t bufLen)
  {
    const Entry* e = find(name);
    if (!e || !itemOf(e->inputs, index, buf, bufLen, nullptr)) return MLGPU_ERR_RANGE;
    return MLGPU_OK;
  }

  int mlgpu_registry_param_name(const char* name, int index, char* buf, size_t bufLen)
  {
    const Entry* e = find(name);
    if (!e || !itemOf(e->params, index, buf, bufLen, nullptr)) return MLGPU_ERR_RANGE;
    return MLGPU_OK;
  }

  int mlgpu_registry_param_index(const char* name, const char* param)
  {
    const Entry* e = find(name);
    if (!e || !param) return -MLGPU_ERR_RANGE;
    char buf[64];
    for (int i = 0; itemOf(e->params, i, buf, sizeof(buf), nullptr); ++i)
      if (!strcmp(buf, param)) return i;
    return -MLGPU_ERR_RANGE;
  }

  // a node by registry name; its inputs by the names of nodes already in the graph
  int mlgpu_graph_add_named(mlgpu_graph* g, const char* procName, const char* nodeName, const char* const* inputNodeNames, int nInputs)
  {
    if (!g) return -MLGPU_ERR_INVALID;
    const Entry* e = find(procName);
    if (!e) return -MLGPU_ERR_RANGE;
    mlgpu_registry_entry d;
    fill(*e, &d);
    if (nInputs < d.n_required_inputs || nInputs > d.n_inputs || nInputs > 8) return -MLGPU_ERR_INVALID;
    int ids[8];
    for (int i = 0; i < nInputs; ++i)
    {
      ids[i] = mlgpu_graph_node(g, inputNodeNames ? inputNodeNames[i] : nullptr);
      if (ids[i] < 0) return ids[i];
    }
    switch (e->nodeType)
    {
      case MLGPU_REGISTRY_OP: return mlgpu_graph_add_op(g, e->kind, ids, nInputs, nodeName);
      case MLGPU_REGISTRY_VOP: return mlgpu_graph_add_vop(g, e->kind, ids, nInputs, nodeName);
      default: return mlgpu_graph_add_proc(g, e->kind, ids, nInputs, nodeName);
    }
  }

  // `filter.coeffs.g1 = ...` by names: the node by its name, the coefficient by the name its registry entry gives it
  int mlgpu_graph_set_named_coeff(mlgpu_graph* g, const char* nodeName, const char* coeffName, const float* perVoice, float uniform)
  {
    if (!g) return MLGPU_ERR_INVALID;
    const int node = mlgpu_graph_node(g, nodeName);
    if (node < 0) return -node;
    const int kind = mlgpu_graph_node_kind(g, node);
    if (kind < 0) return -kind;
    for (const Entry& e : kEntries)
      if (e.nodeType == MLGPU_REGISTRY_PROC && e.kind == kind)
      {
        const int idx = mlgpu_registry_param_index(e.name, coeffName);
        if (idx < 0) return -idx;
        return perVoice ? mlgpu_graph_set_coeff(g, node, idx, perVoice) : mlgpu_graph_set_coeff_uniform(g, node, idx, uniform);
      }
    return MLGPU_ERR_RANGE;
  }

  // the flat per-voice parameter table: a param node by its name
  int mlgpu_graph_set_param_by_name(mlgpu_graph* g, const char* paramName, const float* perVoice, float uniform)
  {
    if (!g) return MLGPU_ERR_INVALID;
    const int node = mlgpu_graph_node(g, paramName);
    if (node < 0) return -node;
    return perVoice ? mlgpu_graph_set_param(g, node, perVoice) : mlgpu_graph_set_param_uniform(g, node, uniform);
  }
}
