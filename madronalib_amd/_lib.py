"""Locate, (re)build and load the C-ABI library madronalib_amd/csrc/libmlgpu.so.

The library is the product: if it is missing and cannot be built, importing any compute
entry fails loudly — there is no Python/CPU fallback.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libmlgpu.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "mlgpu.h")

def _sources():
    """Every file libmlgpu.so is made of: the Makefile's own SRCS and HDRS (so a file added there is never missing here),
    plus the Makefile and the script that embeds the device headers for hiprtc."""
    names = {"Makefile", "embed.py"}
    try:
        text = open(os.path.join(CSRC, "Makefile")).read().replace("\\\n", " ")
        vars_ = {}
        for line in text.splitlines():
            if "=" in line and not line.startswith("\t") and not line.lstrip().startswith("#"):
                k, _, v = line.partition("=")
                vars_[k.strip().rstrip("?:+").strip()] = v.strip()
        for key in ("SRCS", "DEVHDRS", "HDRS"):
            for tok in vars_.get(key, "").split():
                if tok.startswith("$("):
                    continue
                names.add(tok)
    except OSError:
        pass
    return sorted(names)


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(CSRC, s) for s in _sources()] + [HEADER]
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of libmlgpu.so (cross-compiles without a GPU)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        raise RuntimeError(f"hipcc not found at {hipcc}; cannot build {LIB_PATH}")
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j4", f"HIPCC={hipcc}", "libmlgpu.so"], stdout=out)
    return LIB_PATH


_lib = None


def load():
    """Return the ctypes handle to libmlgpu.so, building it first if the sources are newer."""
    global _lib
    if _lib is not None:
        return _lib
    alt = os.environ.get("MLGPU_LIB")   # an alternative build of the same ABI (A/B measurements of a kernel variant)
    if alt:
        _lib = ctypes.CDLL(alt)
        _declare(_lib)
        return _lib
    if needs_build():
        if os.path.exists("/opt/rocm/bin/hipcc") or os.environ.get("HIPCC"):
            build()
        elif not os.path.exists(LIB_PATH):
            raise RuntimeError("libmlgpu.so is not built and hipcc is unavailable; run __graft_entry__.build()")
    _lib = ctypes.CDLL(LIB_PATH)
    _declare(_lib)
    return _lib


def _declare(L):
    c = ctypes
    vp, sz, i, f = c.c_void_p, c.c_size_t, c.c_int, c.c_float
    fp = c.POINTER(c.c_float)
    pp = c.POINTER(c.c_void_p)

    def sig(name, res, args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args

    sig("mlgpu_abi_version", i, [])
    sig("mlgpu_make_window", i, [vp, c.c_size_t, i])
    sig("mlgpu_device_source_hash", c.c_char_p, [])
    sig("mlgpu_status_string", c.c_char_p, [i])
    sig("mlgpu_device_count", i, [])
    sig("mlgpu_device_info", i, [i, c.c_char_p, sz, c.POINTER(i), c.POINTER(c.c_uint64)])
    sig("mlgpu_device_pci_bus_id", i, [i, c.c_char_p, sz])
    sig("mlgpu_device_synchronize", i, [i])
    sig("mlgpu_engine_create", i, [i, pp])
    sig("mlgpu_engine_create_urgency", i, [i, i, pp])
    sig("mlgpu_engine_set_flush_denormals", i, [vp, i])
    sig("mlgpu_engine_get_flush_denormals", i, [vp])
    sig("mlgpu_fence_create", i, [vp, c.POINTER(vp)])
    sig("mlgpu_fence_destroy", i, [vp])
    sig("mlgpu_engine_signal", i, [vp, vp])
    sig("mlgpu_engine_wait", i, [vp, vp])
    sig("mlgpu_engine_set_strict_svf", i, [vp, i])
    sig("mlgpu_engine_get_strict_svf", i, [vp])
    sig("mlgpu_engine_set_cascade_lanes", i, [vp, i])
    sig("mlgpu_engine_get_cascade_lanes", i, [vp])
    sig("mlgpu_engine_create_on_stream", i, [i, vp, pp])
    sig("mlgpu_engine_destroy", i, [vp])
    sig("mlgpu_engine_sync", i, [vp])
    sig("mlgpu_engine_stream", vp, [vp])
    sig("mlgpu_engine_device", i, [vp])
    sig("mlgpu_last_error", c.c_char_p, [vp])
    sig("mlgpu_validate", i, [vp, vp, sz, c.POINTER(c.c_uint64), c.POINTER(c.c_uint64)])
    sig("mlgpu_jit_stats", i, [c.POINTER(c.c_uint64)] * 3 + [c.POINTER(c.c_double)] * 2)
    sig("mlgpu_jit_cache_export", i, [vp, sz, c.POINTER(c.c_size_t)])
    sig("mlgpu_jit_cache_import", i, [vp, sz, c.POINTER(c.c_size_t)])
    sig("mlgpu_jit_cache_clear_memory", i, [])
    sig("mlgpu_jit_compiler_available", i, [])
    sig("mlgpu_registry_count", i, [])
    sig("mlgpu_registry_get", i, [i, vp])
    sig("mlgpu_registry_lookup", i, [c.c_char_p, vp])
    sig("mlgpu_registry_input_name", i, [c.c_char_p, i, c.c_char_p, sz])
    sig("mlgpu_registry_param_name", i, [c.c_char_p, i, c.c_char_p, sz])
    sig("mlgpu_registry_param_index", i, [c.c_char_p, c.c_char_p])
    sig("mlgpu_graph_add_named", i, [vp, c.c_char_p, c.c_char_p, c.POINTER(c.c_char_p), i])
    sig("mlgpu_graph_set_named_coeff", i, [vp, c.c_char_p, c.c_char_p, vp, f])
    sig("mlgpu_graph_set_param_by_name", i, [vp, c.c_char_p, vp, f])
    sig("mlgpu_graph_node_kind", i, [vp, i])
    sig("mlgpu_graph_set_node_name", i, [vp, i, c.c_char_p])
    sig("mlgpu_alloc", i, [vp, sz, pp])
    sig("mlgpu_free", i, [vp, vp])
    sig("mlgpu_upload", i, [vp, vp, vp, sz])
    sig("mlgpu_download", i, [vp, vp, vp, sz])
    sig("mlgpu_fill32", i, [vp, vp, c.c_uint32, sz])
    sig("mlgpu_timer_start", i, [vp])
    sig("mlgpu_timer_stop_ms", i, [vp, fp])
    sig("mlgpu_timer_laps_begin", i, [vp, sz])
    sig("mlgpu_timer_lap", i, [vp])
    sig("mlgpu_timer_laps_end", i, [vp, fp, sz, c.POINTER(c.c_size_t)])
    sig("mlgpu_op_apply", i, [vp, i, vp, vp, vp, vp, sz])
    sig("mlgpu_op_apply_rows1", i, [vp, i, vp, vp, vp, sz])
    sig("mlgpu_row_reduce", i, [vp, i, vp, vp, sz])
    sig("mlgpu_layout_convert", i, [vp, vp, i, vp, i, sz, sz])
    lg = c.c_long
    sig("mlgpu_rows_map", i, [vp, i, lg, lg, i, vp, sz, vp, sz, sz, sz, sz, sz])
    sig("mlgpu_rows_add", i, [vp, vp, sz, vp, sz])
    sig("mlgpu_rows_normalize", i, [vp, vp, vp, sz])
    sig("mlgpu_rows_index", i, [vp, vp, sz, sz])
    sig("mlgpu_multiplex", i, [vp, vp, sz, pp, i, vp, sz, i])
    sig("mlgpu_demultiplex", i, [vp, vp, sz, vp, pp, i, sz, i])
    sig("mlgpu_graph_add_route", i, [vp, i, c.POINTER(c.c_int), i, i, i, c.c_char_p])
    sig("mlgpu_bank_create", i, [vp, c.POINTER(c.c_int32), i, sz, pp])
    sig("mlgpu_bank_destroy", i, [vp])
    sig("mlgpu_bank_num_voices", sz, [vp])
    sig("mlgpu_bank_num_procs", i, [vp])
    sig("mlgpu_bank_num_coeffs", i, [vp, i])
    sig("mlgpu_bank_num_state", i, [vp, i])
    sig("mlgpu_bank_clear", i, [vp])
    sig("mlgpu_bank_set_coeff", i, [vp, i, i, vp])
    sig("mlgpu_bank_get_coeff", i, [vp, i, i, vp])
    sig("mlgpu_bank_set_coeff_uniform", i, [vp, i, i, f])
    sig("mlgpu_bank_get_state", i, [vp, i, i, vp])
    sig("mlgpu_bank_set_state", i, [vp, i, i, vp])
    sig("mlgpu_bank_set_state_uniform", i, [vp, i, i, c.c_uint32])
    sig("mlgpu_bank_set_input_const", i, [vp, vp])
    sig("mlgpu_bank_process", i, [vp, sz, vp, i, vp, i])
    sig("mlgpu_bank_process_mixdown", i, [vp, sz, vp, i, vp, vp])
    sig("mlgpu_bank_prepare_mixdown", i, [vp])
    sig("mlgpu_bank_is_fused", i, [vp])
    sig("mlgpu_bank_kernel_name", c.c_char_p, [vp])
    ip = c.POINTER(c.c_int)
    sig("mlgpu_engine_set_jit", i, [vp, i])
    sig("mlgpu_jit_selftest", i, [c.c_char_p, sz])
    sig("mlgpu_graph_create", i, [vp, sz, pp])
    sig("mlgpu_graph_destroy", i, [vp])
    sig("mlgpu_graph_add_input", i, [vp, c.c_char_p])
    sig("mlgpu_graph_add_param", i, [vp, c.c_char_p])
    sig("mlgpu_graph_add_const", i, [vp, f])
    sig("mlgpu_graph_set_live_constants", i, [vp, i])
    sig("mlgpu_graph_set_const", i, [vp, i, f])
    sig("mlgpu_graph_update_constants_from", i, [vp, vp])
    sig("mlgpu_graph_add_const_vector", i, [vp, ctypes.POINTER(ctypes.c_float), ctypes.c_char_p])
    sig("mlgpu_graph_add_control", i, [vp, c.c_char_p])
    sig("mlgpu_graph_add_feedback", i, [vp, c.c_char_p])
    sig("mlgpu_graph_set_feedback", i, [vp, i, i])
    sig("mlgpu_graph_set_max_delay", i, [vp, i, f])
    sig("mlgpu_allpass1_make_coeffs", f, [f])
    sig("mlgpu_fractional_delay_make_state", None, [f, fp])
    sig("mlgpu_graph_add_vop", i, [vp, i, ip, i, c.c_char_p])
    sig("mlgpu_graph_process_ctl", i, [vp, sz, pp, i, pp, pp, i])
    sig("mlgpu_graph_last_error", c.c_char_p, [vp])
    sig("mlgpu_graph_set_output_group_sum", i, [vp, i, i])
    sig("mlgpu_graph_set_output_mixdown", i, [vp, i, i])
    sig("mlgpu_graph_reserve_mixdown", i, [vp, sz])
    sig("mlgpu_graph_set_input_group", i, [vp, i, i])
    sig("mlgpu_graph_add_event_row", i, [vp, i, c.c_char_p])
    sig("mlgpu_graph_bind_events", i, [vp, vp])
    sig("mlgpu_graph_process_events", i, [vp, sz, i, pp, i, pp, pp, i])
    sig("mlgpu_linear_glide_make_coeffs", None, [f, fp])
    sig("mlgpu_sample_accurate_linear_glide_make_coeffs", None, [f, fp])
    sig("mlgpu_graph_add_proc", i, [vp, i, ip, i, c.c_char_p])
    sig("mlgpu_graph_add_op", i, [vp, i, ip, i, c.c_char_p])
    sig("mlgpu_graph_add_output", i, [vp, i])
    sig("mlgpu_graph_node", i, [vp, c.c_char_p])
    sig("mlgpu_graph_num_nodes", i, [vp])
    sig("mlgpu_graph_compile", i, [vp])
    sig("mlgpu_graph_compile_async", i, [vp])
    sig("mlgpu_graph_compile_poll", i, [vp])
    sig("mlgpu_behaviour_revision", i, [])
    sig("mlgpu_graph_source", c.c_char_p, [vp])
    sig("mlgpu_graph_node_use_count", i, [vp, i])
    sig("mlgpu_graph_emit", i, [vp, c.POINTER(vp), c.POINTER(c.c_size_t)])
    sig("mlgpu_graph_begin_region", i, [vp, i, c.POINTER(i), i, c.POINTER(i)])
    sig("mlgpu_graph_end_region", i, [vp, i, c.c_char_p])
    sig("mlgpu_graph_clear", i, [vp])
    sig("mlgpu_graph_clear_proc", i, [vp, i])
    sig("mlgpu_graph_set_input_layout", i, [vp, i, i])
    sig("mlgpu_graph_set_voices_per_lane", i, [vp, i])
    sig("mlgpu_graph_set_delay_layout", i, [vp, i])
    sig("mlgpu_graph_delay_layout", i, [vp])
    sig("mlgpu_graph_device_bytes", sz, [vp])
    sig("mlgpu_graph_set_autotune", i, [vp, i])
    sig("mlgpu_graph_tuning", i, [vp, c.POINTER(i), c.POINTER(i)])
    sig("mlgpu_graph_workgroups_per_cu", i, [vp])
    sig("mlgpu_mixdown_reserve", i, [vp, sz, sz])
    sig("mlgpu_mixdown_shard_level", i, [sz])
    sig("mlgpu_mixdown_shard_rows", sz, [sz])
    sig("mlgpu_mixdown_shard", i, [vp, vp, i, sz, sz, vp, vp])
    sig("mlgpu_bank_process_mixdown_shard", i, [vp, sz, vp, i, vp, vp])
    sig("mlgpu_mixdown_finish", i, [fp, sz, sz, i, fp, fp])
    sig("mlgpu_mixdown", i, [vp, vp, i, sz, sz, vp, vp])
    sig("mlgpu_mixdown_groups", i, [vp, vp, i, sz, sz, sz, vp, i])
    sig("mlgpu_events_create", i, [vp, sz, i, pp])
    sig("mlgpu_events_destroy", i, [vp])
    sig("mlgpu_events_clear", i, [vp])
    sig("mlgpu_events_set_sample_rate", i, [vp, c.c_double])
    sig("mlgpu_events_set_protocol", i, [vp, i])
    sig("mlgpu_events_set_unison", i, [vp, i])
    sig("mlgpu_events_set_mod_cc", i, [vp, i])
    sig("mlgpu_events_set_pitch_bend_semitones", i, [vp, f])
    sig("mlgpu_events_set_mpe_pitch_bend_semitones", i, [vp, f])
    sig("mlgpu_events_set_pitch_glide_seconds", i, [vp, f])
    sig("mlgpu_events_set_drift_amount", i, [vp, f])
    sig("mlgpu_events_set_wanted_rows", i, [vp, c.c_uint])
    sig("mlgpu_events_num_voices", sz, [vp])
    sig("mlgpu_events_newest_voice", i, [vp, sz])
    sig("mlgpu_events_add_event", i, [vp, sz, vp])
    sig("mlgpu_events_add_events", i, [vp, vp, vp, sz])
    sig("mlgpu_events_clear_events", i, [vp])
    sig("mlgpu_events_process", i, [vp, sz, i, pp, i])
    sig("mlgpu_transport_create", i, [vp, sz, sz, ctypes.POINTER(vp)])
    sig("mlgpu_transport_destroy", i, [vp])
    sig("mlgpu_transport_reserve", i, [vp, sz])
    sig("mlgpu_transport_clear", i, [vp, sz])
    sig("mlgpu_transport_set_time_and_rate", i, [vp, sz, ctypes.c_double, ctypes.c_double, i, ctypes.c_double])
    sig("mlgpu_transport_process", i, [vp, sz])
    sig("mlgpu_transport_beat_phase", vp, [vp])
    sig("mlgpu_transport_samples_since_start", ctypes.c_uint64, [vp, sz])
    sig("mlgpu_transport_bpm", ctypes.c_double, [vp, sz])
    sig("mlgpu_events_watch_controllers", i, [vp, ctypes.POINTER(ctypes.c_int), i, sz])
    sig("mlgpu_events_reserve_for_graph", i, [vp, sz])
    sig("mlgpu_events_graph_reserve_bytes", sz, [vp, sz])
    sig("mlgpu_events_controller_signal", vp, [vp, i])
    sig("mlgpu_resampler_create", i, [vp, sz, i, i, pp])
    sig("mlgpu_resampler_destroy", i, [vp])
    sig("mlgpu_resampler_clear", i, [vp])
    sig("mlgpu_resampler_get_state", i, [vp, vp])
    sig("mlgpu_resampler_set_state", i, [vp, vp])
    sig("mlgpu_resampler_process", i, [vp, sz, vp, i, vp, i])
    sig("mlgpu_engine_begin_recording", i, [vp])
    sig("mlgpu_engine_end_recording", i, [vp, pp])
    sig("mlgpu_sequence_launch", i, [vp])
    sig("mlgpu_sequence_num_nodes", sz, [vp])
    sig("mlgpu_sequence_destroy", i, [vp])
    sig("mlgpu_published_signal_create", i, [vp, i, i, i, i, pp])
    sig("mlgpu_published_signal_destroy", i, [vp])
    sig("mlgpu_published_signal_write", i, [vp, sz, pp, i, sz, sz, sz])
    sig("mlgpu_published_signal_num_channels", sz, [vp])
    sig("mlgpu_published_signal_read_available", sz, [vp])
    sig("mlgpu_published_signal_available_frames", sz, [vp])
    sig("mlgpu_published_signal_read", sz, [vp, vp, sz])
    sig("mlgpu_published_signal_read_latest", sz, [vp, vp, sz])
    sig("mlgpu_published_signal_peek_latest", None, [vp, vp, sz])
    sig("mlgpu_dspbuffer_create", vp, [])
    sig("mlgpu_dspbuffer_destroy", None, [vp])
    sig("mlgpu_dspbuffer_resize", sz, [vp, i])
    sig("mlgpu_dspbuffer_size", sz, [vp])
    sig("mlgpu_dspbuffer_clear", None, [vp])
    sig("mlgpu_dspbuffer_read_available", sz, [vp])
    sig("mlgpu_dspbuffer_write_available", sz, [vp])
    sig("mlgpu_dspbuffer_write", None, [vp, vp, sz])
    sig("mlgpu_dspbuffer_read", sz, [vp, vp, sz])
    sig("mlgpu_dspbuffer_read_vector", i, [vp, vp])
    sig("mlgpu_dspbuffer_discard", None, [vp, sz])
    sig("mlgpu_dspbuffer_write_with_overlap_add", None, [vp, vp, sz, sz])
    sig("mlgpu_dspbuffer_read_with_overlap", None, [vp, vp, sz, sz])
    sig("mlgpu_dspbuffer_peek_most_recent", None, [vp, vp, sz])
    sig("mlgpu_process_buffer_create", i, [vp, sz, sz, sz, pp])
    sig("mlgpu_process_buffer_destroy", i, [vp])
    sig("mlgpu_process_buffer_set_pipelined", i, [vp, i])
    sig("mlgpu_process_buffer_latency_frames", sz, [vp])
    sig("mlgpu_process_buffer_process", i, [vp, pp, pp, i, vp, vp])
    sig("mlgpu_graph_set_state_uniform", i, [vp, i, i, c.c_uint32])
    sig("mlgpu_graph_set_param", i, [vp, i, vp])
    sig("mlgpu_graph_set_param_uniform", i, [vp, i, f])
    sig("mlgpu_graph_num_coeffs", i, [vp, i])
    sig("mlgpu_graph_num_state", i, [vp, i])
    sig("mlgpu_graph_set_coeff", i, [vp, i, i, vp])
    sig("mlgpu_graph_set_coeff_uniform", i, [vp, i, i, f])
    sig("mlgpu_graph_get_state", i, [vp, i, i, vp])
    sig("mlgpu_graph_set_state", i, [vp, i, i, vp])
    sig("mlgpu_graph_process", i, [vp, sz, pp, i, pp, i])
    sig("mlgpu_lopass_make_coeffs", None, [f, f, fp])
    sig("mlgpu_hipass_make_coeffs", None, [f, f, fp])
    sig("mlgpu_bandpass_make_coeffs", None, [f, f, fp])
    sig("mlgpu_loshelf_make_coeffs", None, [f, f, f, fp])
    sig("mlgpu_hishelf_make_coeffs", None, [f, f, f, fp])
    sig("mlgpu_bell_make_coeffs", None, [f, f, f, fp])
    sig("mlgpu_onepole_make_coeffs", None, [f, fp])
    sig("mlgpu_dcblocker_make_coeffs", f, [f])
    sig("mlgpu_adsr_calc_coeffs", None, [f, f, f, f, f, fp])
    sig("mlgpu_db_to_gain", f, [f])
