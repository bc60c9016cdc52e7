"""Host-side mirror of the reference's Seq driver (py/sequence.py) on top of the C-ABI.

Names and argument meaning follow the reference so tests read like its code:

    reference (py/sequence.py)                      here
    ------------------------------------------      -------------------------------------------
    get_reference_seqs(paths)            :189-194   Sequencer.get_reference_seqs(paths)
    mdf_generator(f)                     :197-221   Sequencer.batch_from_mdf(text) (parsed in C++)
    mdf_to_seq(mdf, targets)             :303-320   Sequencer.mdf_to_seq(molecules, target, ...)
    perfect / badread                    :242-270   target="perfect" / "badread"
    fastq_formatter / fasta_formatter    :273-288   fastq=True / False
    Identities / ErrorModel / QScoreModel           set_identity / load_error_model / load_qscore_model

All arithmetic happens in libtksmseq.so (HIP kernels); nothing here computes sequence data.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class TksmSeqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"tksmseq error {code}: {msg}")
        self.code = code


class Batch:
    def __init__(self, seq, handle):
        self._seq, self._h = seq, handle
        n, ni, nm = C.c_uint64(), C.c_uint64(), C.c_uint64()
        seq._lib.tksmseq_batch_info(handle, C.byref(n), C.byref(ni), C.byref(nm))
        self.n_reads, self.n_intervals, self.n_mods = n.value, ni.value, nm.value

    def free(self):
        if self._h:
            self._seq._lib.tksmseq_batch_free(self._seq._ctx, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class RunResult:
    def __init__(self, seq, res):
        self._seq = seq
        self.records_ptr, self.offsets_ptr = res.records, res.record_offsets
        self.records_bytes, self.n_reads = res.records_bytes, res.n_reads
        self.bases_in, self.bases_out = res.bases_in, res.bases_out
        self.kernel_ms = list(res.kernel_ms)

    def download(self):
        rec = np.empty(self.records_bytes, np.uint8)
        off = np.empty(self.n_reads + 1, np.uint64)
        self._seq._chk(self._seq._lib.tksmseq_result_download(self._seq._ctx, rec.ctypes.data, off.ctypes.data))
        return rec.tobytes(), off

    def download_range(self, offset, nbytes):
        """bytes [offset, offset + nbytes) of the record stream (tksmseq_result_download_range)"""
        rec = np.empty(nbytes, np.uint8)
        self._seq._chk(self._seq._lib.tksmseq_result_download_range(self._seq._ctx, rec.ctypes.data, offset, nbytes, 0))
        return rec.tobytes()

    def copy_to_device(self, records_ptr=None, offsets_ptr=None):
        self._seq._chk(self._seq._lib.tksmseq_result_copy_device(
            self._seq._ctx, C.c_void_p(records_ptr) if records_ptr else None, C.c_void_p(offsets_ptr) if offsets_ptr else None))

    def records(self):
        rec, off = self.download()
        return [rec[int(off[i]):int(off[i + 1])] for i in range(self.n_reads)]

    def stats(self):
        ist = np.empty((self.n_reads, 16), np.int32)
        dst = np.empty((self.n_reads, 2), np.float64)
        self._seq._chk(self._seq._lib.tksmseq_stats_download(self._seq._ctx, ist.ctypes.data, dst.ctypes.data))
        return ist, dst


class Sequencer:
    """One context = one GPU (the reference's module globals: reference_seqs, identities, models).

    Several contexts in flight (clone(), one host thread and stream each) want a hardware queue each: start the process with
    GPU_MAX_HW_QUEUES=16 in its environment (read once, when the HIP runtime starts; its default of 4 makes the streams of different
    contexts share queues, and their kernels then run one after the other).  The `tksm` CLI and bench.py set it for their own
    processes; importing this module does not touch the embedding application's environment."""

    def __init__(self, device=0, stream=None):
        self._lib = L.load()
        ctx = C.c_void_p()
        rc = self._lib.tksmseq_create(device, C.byref(ctx))
        if rc:
            raise TksmSeqError(rc, self._lib.tksmseq_last_error(None).decode())
        self._ctx = ctx
        if stream is not None:
            self._chk(self._lib.tksmseq_set_stream(self._ctx, C.c_void_p(stream)))

    def clone(self, stream=None):
        """A second Sequencer on the same device that shares this one's packed reference and model tables (for another
        host thread / batch in flight).  Close clones before the source.  (Hardware queues: GPU_MAX_HW_QUEUES=16, see the class.)"""
        other = object.__new__(Sequencer)
        other._lib = self._lib
        ctx = C.c_void_p()
        self._chk(self._lib.tksmseq_clone(self._ctx, C.byref(ctx)))
        other._ctx = ctx
        other._parent = self          # keeps the source alive
        if stream is not None:
            other._chk(self._lib.tksmseq_set_stream(other._ctx, C.c_void_p(stream)))
        return other

    def close(self):
        if self._ctx:
            self._lib.tksmseq_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise TksmSeqError(rc, self._lib.tksmseq_last_error(self._ctx).decode())

    # ---- reference
    def get_reference_seqs(self, paths):
        for p in paths:
            self._chk(self._lib.tksmseq_reference_add_fasta(self._ctx, str(p).encode()))

    def add_contig(self, name, seq):
        """seq: bytes/str (host) or an object with data_ptr()/numel() holding ASCII bytes on this GPU."""
        if hasattr(seq, "data_ptr"):
            self._chk(self._lib.tksmseq_reference_add_contig(self._ctx, name.encode(), C.c_void_p(seq.data_ptr()),
                                                             seq.numel(), 1))
        else:
            b = seq.encode() if isinstance(seq, str) else bytes(seq)
            buf = (C.c_char * len(b)).from_buffer_copy(b) if len(b) else None
            self._chk(self._lib.tksmseq_reference_add_contig(self._ctx, name.encode(),
                                                             C.cast(buf, C.c_void_p) if buf is not None else None, len(b), 0))

    def contig_id(self, name):
        return self._lib.tksmseq_reference_contig_id(self._ctx, name.encode())

    def reference_info(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self._lib.tksmseq_reference_info(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return {"n_contigs": a.value, "total_bases": b.value, "device_bytes": c.value}

    # ---- models
    def load_error_model(self, name_or_path):
        self._chk(self._lib.tksmseq_load_error_model(self._ctx, str(name_or_path).encode()))

    def load_qscore_model(self, name_or_path):
        self._chk(self._lib.tksmseq_load_qscore_model(self._ctx, str(name_or_path).encode()))

    def load_tail_model(self, name_or_path="no_noise"):
        """KDE_noise_generator.load (py/tksm_badread.py:944-962); "no_noise" switches the tail off."""
        self._chk(self._lib.tksmseq_load_tail_model(self._ctx, str(name_or_path).encode()))

    def set_tail_model(self, lx, ly, grid, trans, ratio, bases="AGTC"):
        """The same from arrays (KDE_noise_generator.__init__, py/tksm_badread.py:905-917)."""
        lx = np.ascontiguousarray(lx, np.float64); ly = np.ascontiguousarray(ly, np.float64)
        grid = np.ascontiguousarray(grid, np.float64); trans = np.ascontiguousarray(trans, np.float64)
        if grid.shape != (len(ly), len(lx)) or trans.shape != (4, 4) or len(bases) != 4:
            raise ValueError("tail model: grid must be len(ly) x len(lx), trans 4 x 4, bases 4 symbols")
        b = bases.encode() if isinstance(bases, str) else bytes(bases)
        d = L.TailModelDesc(len(lx), len(ly), lx.ctypes.data, ly.ctypes.data, grid.ctypes.data, (C.c_double * 16)(*trans.ravel()),
                          float(ratio), (C.c_uint8 * 4)(*b), (C.c_uint8 * 4)())
        self._chk(self._lib.tksmseq_set_tail_model(self._ctx, C.byref(d)))

    def set_identity(self, mean=84.0, max_identity=99.0, stdev=5.5):
        self._chk(self._lib.tksmseq_set_identity(self._ctx, mean, max_identity, stdev))

    def error_model_tables(self):
        t, k, a = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._lib.tksmseq_get_error_model(self._ctx, C.byref(t), C.byref(k), C.byref(a), None, None, None))
        n = 4 ** k.value
        cdf, alts, nalts = np.empty((n, a.value), np.uint32), np.empty((n, a.value), np.uint64), np.empty(n, np.uint8)
        self._chk(self._lib.tksmseq_get_error_model(self._ctx, None, None, None, cdf.ctypes.data, alts.ctypes.data,
                                                    nalts.ctypes.data))
        return {"type": t.value, "k": k.value, "max_alts": a.value, "cdf": cdf, "alts": alts, "nalts": nalts}

    def qscore_model_tables(self):
        ns, ks, pl = C.c_int32(), C.c_int32(), C.c_uint64()
        self._chk(self._lib.tksmseq_get_qscore_model(self._ctx, C.byref(ns), C.byref(ks), C.byref(pl), None, None, None,
                                                     None, None))
        keys, off, cnt = np.empty(ns.value, np.uint64), np.empty(ns.value, np.uint32), np.empty(ns.value, np.uint32)
        cdf, q = np.empty(pl.value, np.uint32), np.empty(pl.value, np.uint8)
        self._chk(self._lib.tksmseq_get_qscore_model(self._ctx, None, None, None, keys.ctypes.data, off.ctypes.data,
                                                     cnt.ctypes.data, cdf.ctypes.data, q.ctypes.data))
        return {"n_slots": ns.value, "kmer_size": ks.value, "keys": keys, "row_off": off, "row_cnt": cnt,
                "cdf_pool": cdf, "q_pool": q}

    def identity_tables(self):
        c, v, a, b = C.c_int32(), C.c_double(), C.c_double(), C.c_double()
        self._chk(self._lib.tksmseq_get_identity(self._ctx, C.byref(c), C.byref(v), C.byref(a), C.byref(b), None))
        qtab = None
        if not c.value:
            qtab = np.empty(65537, np.float64)
            self._chk(self._lib.tksmseq_get_identity(self._ctx, None, None, None, None, qtab.ctypes.data))
        return {"constant": bool(c.value), "value": v.value, "beta_a": a.value, "beta_b": b.value, "qtab": qtab}

    # ---- batches
    def batch_from_mdf(self, text):
        b = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        self._chk(self._lib.tksmseq_batch_from_mdf_text(self._ctx, b, len(b), C.byref(h)))
        return Batch(self, h)

    def batch_from_arrays(self, reads, intervals, mods=None, literals=None, literal_pool=b"", ids=None, id_pool=b""):
        """Binary layout of include/tksmseq.h (numpy arrays)."""
        reads = np.ascontiguousarray(reads, np.uint32).reshape(-1, 2)
        intervals = np.ascontiguousarray(intervals, np.uint32).reshape(-1, 4)
        mods = np.ascontiguousarray(mods if mods is not None else np.zeros((0, 2)), np.uint32).reshape(-1, 2)
        literals = np.ascontiguousarray(literals if literals is not None else np.zeros((0, 2)), np.uint64).reshape(-1, 2)
        if ids is None:
            ids = np.zeros((len(reads), 2), np.uint32)
        ids = np.ascontiguousarray(ids, np.uint32).reshape(-1, 2)
        lp = np.frombuffer(bytes(literal_pool), np.uint8) if not isinstance(literal_pool, np.ndarray) else literal_pool
        ip = np.frombuffer(bytes(id_pool), np.uint8) if not isinstance(id_pool, np.ndarray) else id_pool
        lp, ip = np.ascontiguousarray(lp, np.uint8), np.ascontiguousarray(ip, np.uint8)
        d = L.BatchDesc(len(reads), len(intervals), len(mods), len(literals), len(lp), len(ip), reads.ctypes.data,
                        intervals.ctypes.data, mods.ctypes.data, literals.ctypes.data, lp.ctypes.data, ids.ctypes.data,
                        ip.ctypes.data)
        h = C.c_void_p()
        self._chk(self._lib.tksmseq_batch_create(self._ctx, C.byref(d), C.byref(h)))
        return Batch(self, h)

    # ---- molecule-description transforms upstream of Seq (device to device)
    def pcr(self, batch, cycles, target_count, error_rate=None, efficiency=None, preset=None, seed=42, templates=None):
        """PCR::perform (src/pcr.cpp:66-89) on the device; preset: one of the names of src/pcr.cpp:136-140; templates = (begin,
        end): the copies of that slice of the input molecules only (slices, one after the other = the whole)."""
        if preset is not None:
            er, ef = C.c_double(), C.c_double()
            if self._lib.tksmseq_pcr_preset(preset.encode(), C.byref(er), C.byref(ef)):
                raise ValueError(f"Preset {preset} not found")
            error_rate = er.value if error_rate is None else error_rate
            efficiency = ef.value if efficiency is None else efficiency
        if error_rate is None or efficiency is None:
            raise ValueError("Error rate is required!" if error_rate is None else "Efficiency is required!")
        tb, te = templates if templates is not None else (0, 0)
        p = L.PcrParams(seed, target_count, cycles, 0, error_rate, efficiency, tb, te)
        h = C.c_void_p()
        self._chk(self._lib.tksmseq_pcr(self._ctx, batch._h, C.byref(p), C.byref(h)))
        return Batch(self, h)

    def pcr_template_counts(self, batch, cycles, target_count, error_rate, efficiency, seed=42):
        """written copies per input molecule (tksmseq_pcr_template_counts)"""
        import numpy as np
        p = L.PcrParams(seed, target_count, cycles, 0, error_rate, efficiency, 0, 0)
        out = np.zeros(batch.n_reads, np.uint64)
        self._chk(self._lib.tksmseq_pcr_template_counts(self._ctx, batch._h, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def truncate(self, batch, normal=None, lognormal=None, kde_model=None, always_end=False, kde_models_length=False, seed=42,
                 first_molecule_index=0):
        """truncate_transformer / truncate_transformer_kde (src/truncate.cpp:322-351) on the device: exactly one of normal=(mu,
        sigma), lognormal=(mu, sigma), kde_model=path."""
        if (normal is not None) + (lognormal is not None) + (kde_model is not None) != 1:
            raise ValueError("One of kde-model, normal or lognormal is required!" if normal is None and lognormal is None and kde_model is None
                             else "Only one of kde-model, normal or lognormal is allowed!")
        mode = L.TRC_NORMAL if normal is not None else L.TRC_LOGNORMAL if lognormal is not None else L.TRC_KDE
        mu, sigma = normal if normal is not None else lognormal if lognormal is not None else (0.0, 0.0)
        path = str(kde_model).encode() if kde_model is not None else None
        p = L.TrcParams(seed, first_molecule_index, mode, 1 if always_end else 0, 1 if kde_models_length else 0, 0, mu, sigma, path)
        h = C.c_void_p()
        self._chk(self._lib.tksmseq_truncate(self._ctx, batch._h, C.byref(p), C.byref(h)))
        return Batch(self, h)

    def to_mdf_text(self, batch):
        """molecule_descriptor::operator<< of every molecule (src/interval.h:898-905)."""
        t, n = C.c_void_p(), C.c_uint64()
        self._chk(self._lib.tksmseq_batch_to_mdf_text(self._ctx, batch._h, C.byref(t), C.byref(n)))
        try:
            return C.string_at(t, n.value).decode()
        finally:
            self._lib.tksmseq_text_free(t)

    # ---- the hot path
    def run(self, batch, target="badread", fastq=True, compute_qual=True, seed=42, first_read_index=0, stride=1,
            collect_stats=False, perfect_of_badread=False):
        p = L.RunParams(seed, first_read_index, stride, L.MODE_BADREAD if target == "badread" else L.MODE_PERFECT,
                        1 if fastq else 0, 1 if compute_qual else 0, 1 if collect_stats else 0,
                        1 if perfect_of_badread else 0, 0)
        r = L.Result()
        self._chk(self._lib.tksmseq_run(self._ctx, batch._h, C.byref(p), C.byref(r)))
        return RunResult(self, r)

    def mdf_to_seq(self, molecules, target="perfect", **kw):
        """molecules: iterable of (molecule_id, [(chrom, start, end, strand, modifications), ...]) exactly as the
        reference's mdf_generator yields them (depth already unrolled).  Returns the formatted records."""
        lines = []
        for mid, intervals in molecules:
            lines.append(f"+{mid}\t1\t\n")
            for chrom, start, end, strand, mods in intervals:
                lines.append(f"{chrom}\t{start}\t{end}\t{strand}\t{mods}\n")
        b = self.batch_from_mdf("".join(lines))
        try:
            return self.run(b, target=target, **kw).records()
        finally:
            b.free()

    def set_host_threads(self, n):
        """host threads of the MDF parser and writer (tksmseq_set_host_threads; the CLI's -t)"""
        self._chk(self._lib.tksmseq_set_host_threads(self._ctx, int(n)))

    def set_timing(self, on=True):
        self._chk(self._lib.tksmseq_set_timing(self._ctx, 1 if on else 0))

    def run_diagnostics(self):
        """counts of the last Badread run (tksmseq_run_diagnostics): rounds, reads that took the exact kernel, redo share, fall-backs"""
        out = (C.c_uint32 * 16)()
        self._chk(self._lib.tksmseq_run_diagnostics(self._ctx, out))
        keys = ("rounds", "exact_kernel_reads", "predicted_stragglers", "jobs_14_row_rounds", "jobs_redone_full_width", "fallbacks",
                "fallback_reasons", "fallbacks_qscore_jobs", "fallbacks_list_pass", "jobs_all_rounds", "band_exits")
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def set_output_buffer(self, ptr, capacity):
        self._chk(self._lib.tksmseq_set_output_buffer(self._ctx, C.c_void_p(ptr) if ptr else None, capacity))

    def synchronize(self):
        self._chk(self._lib.tksmseq_synchronize(self._ctx))

    def interleave_records(self, streams, offsets, n_per_rank, dst_ptr, dst_capacity):
        n = len(streams)
        sp = (C.c_void_p * n)(*[C.c_void_p(s) for s in streams])
        op = (C.c_void_p * n)(*[C.c_void_p(o) for o in offsets])
        npr = (C.c_uint64 * n)(*n_per_rank)
        out = C.c_uint64()
        self._chk(self._lib.tksmseq_interleave_records(self._ctx, n, sp, op, npr, C.c_void_p(dst_ptr), dst_capacity,
                                                       C.byref(out)))
        return out.value


def sequence_main(argv):
    """The module entry point: `tksm sequence ...` (src/tksm.cpp:164-166); argv[0] == "sequence"."""
    lib = L.load()
    arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    return lib.tksmseq_sequence_main(len(argv), arr)
