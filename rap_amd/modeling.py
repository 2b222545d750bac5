"""Host-side mirror of the sampling boundary of ``rectified_point_flow/modeling.py``.

``RectifiedPointFlow`` here is a plain class (no Lightning): it keeps the reference's
``sample_rectified_flow`` signature and return structure (modeling.py:632-741) plus the
``fit_transformations`` call that ``test_step`` makes right after it (modeling.py:389-391).  The whole
Euler loop -- velocity network, Euler update, optional per-step Procrustes rigidity projection,
trajectory capture, final pose recovery -- is ONE call into librapflow (``rap_sample``); nothing on
the path syncs with the host.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from .flow_model import PointCloudDiT, _f32c, _require_cuda, workspace


def shard_cuts(cu_seqlens_host, n_shards: int) -> list[int]:
    """Sample indices [0, b_1, ..., B] that cut a packed batch (``cu_seqlens_host``: B+1 token offsets) into at most ``n_shards``
    contiguous, non-empty shards of about equal TOKEN count: cut k is the sample boundary closest to k/n of the tokens, leaving at
    least one sample for every later shard; no shard is left without tokens (samples without points join a neighbour)."""
    B = len(cu_seqlens_host) - 1
    n = max(1, min(int(n_shards), B))
    TP = cu_seqlens_host[-1] - cu_seqlens_host[0]
    cuts = [0]
    for k in range(1, n):
        target = cu_seqlens_host[0] + TP * k / n
        cand = range(cuts[-1] + 1, B - (n - 1 - k))
        if len(cand) == 0:
            break
        cuts.append(min(cand, key=lambda i: abs(cu_seqlens_host[i] - target)))
    cuts.append(B)
    # a sample may hold zero points (all parts empty): a shard made only of such samples would be a zero-token rap_sample call,
    # which the library refuses -- merge it into its left neighbour (the last one into its right one)
    keep = [0]
    for c in cuts[1:-1]:
        if cu_seqlens_host[c] > cu_seqlens_host[keep[-1]]:
            keep.append(c)
    if len(keep) > 1 and cu_seqlens_host[B] == cu_seqlens_host[keep[-1]]:
        keep.pop()
    keep.append(B)
    return keep


class RectifiedPointFlow:
    """Inference-side drop-in for the reference LightningModule's sampling API."""

    def __init__(self, flow_model: PointCloudDiT = None, inference_sampling_steps: int = 20,
                 inference_sampler: str = "euler", n_generations: int = 1, rigidity_forcing: bool = False,
                 return_end_point_trajectory: bool = True, encoder_on: bool = False, validate_inputs: bool | None = None,
                 num_streams: int | None = None, graph_replay: bool | None = None, **_ignored):
        if flow_model is None:
            raise ValueError("flow_model is required")            # modeling.py:80-81
        if encoder_on:
            # the PTv3 feature extractor itself is outside SURVEY.md section 8; a flow model built with in_dim > 0 takes the latent
            # features it would produce through sample_rectified_flow(data_dict, latent_features, ...) exactly as the reference passes them
            raise NotImplementedError("encoder_on=True: run the PTv3 encoder yourself and pass its output as latent_features "
                                      "(flow_model in_dim > 0)")
        if inference_sampler != "euler":
            raise ValueError(f"Unknown sampler: {inference_sampler}. Available: ['euler']")   # sampler.py:168-169
        self.flow_model = flow_model
        self.inference_sampling_steps = inference_sampling_steps
        self.inference_sampler = inference_sampler
        self.n_generations = n_generations
        self.rigidity_forcing = rigidity_forcing
        self.return_end_point_trajectory = return_end_point_trajectory
        self.last_poses = None
        # The reference asserts the batch layout in split_parts (utils/point_clouds.py:33-52, 41-44) on every call.  Here:
        #   "deferred" (default; True means the same): one tiny device kernel writes a verdict flag BEFORE the sampling call is
        #       enqueued, the results of a call whose flag is non-zero are overwritten with NaN on the device
        #       (rap_poison_on_flag), and the flag travels to pinned host memory behind an event.  The call itself never waits for
        #       the GPU; ValueError is raised the first time the flag can be read without stalling -- at the next sampling call of
        #       this object, or from check_pending() / synchronize().  (Round 3 read the flag back inside the call: one host sync
        #       per call -- and per shard -- which serialised the ~3 300 enqueues behind the previous call's GPU work.)
        #   "eager": read the flag back before enqueueing (one host sync per call), raise immediately -- the reference's behaviour.
        #   False / RAP_VALIDATE_INPUTS=0: no check (a latency-critical caller that validated its batches itself).
        # Skipped while the stream is being captured into a HIP graph (a capture cannot read back).  In EVERY mode the library works
        # on sanitised segment tables -- the part table clamped to the point count, cu_seqlens clamped to [0, TP] and made
        # non-decreasing on the device (round 5) -- so a malformed batch yields wrong (deferred: NaN) results, never an
        # out-of-bounds access.
        if validate_inputs is None:
            env_v = os.environ.get("RAP_VALIDATE_INPUTS", "1")
            validate_inputs = False if env_v == "0" else ("eager" if env_v == "eager" else "deferred")
        if validate_inputs is True:
            validate_inputs = "deferred"
        if validate_inputs not in (False, "deferred", "eager"):
            raise ValueError(f"validate_inputs must be False, True, 'deferred' or 'eager' (got {validate_inputs!r})")
        self.validate_inputs = validate_inputs
        self._pending: list = []          # deferred verdicts: (event, pinned host int32, description)
        self._pin, self._pin_next = None, 0
        # Concurrent batch shards (opt-in, num_streams > 1): samples are independent, so a batch can be run as contiguous shards on
        # several HIP streams.  Round 3: the shards now really fork -- one event recorded on the caller's stream BEFORE
        # any shard is enqueued, every auxiliary stream waits on that event, shard 0 is enqueued LAST on the caller's stream.  (The
        # round-2 code enqueued shard 0 first and made the auxiliary streams wait for the caller's stream afterwards, i.e. for
        # shard 0: its "2 408 / 2 394 / 2 396 ms on 2 / 3 / 4 streams vs 2 421 on one" measured serialized shards, not overlap.)
        # Re-measured numbers: DESIGN.md section 5.  Default: one stream; RAP_NUM_STREAMS overrides.
        env = os.environ.get("RAP_NUM_STREAMS")
        self.num_streams = int(env) if env else num_streams
        self._aux_streams: dict = {}
        # HIP-graph replay (opt-in, round 4; RAP_GRAPH_REPLAY=1): a sampling call is ~2 900 kernel launches.  Enqueued one by one they
        # cost the host 8-9 ms at demo size and, at full size, 0.2 s (bf16) to 1.6 s (fp32) because the call is a few hundred commands
        # longer than the HIP queue and the host has to follow the GPU; replayed as ONE captured graph the same call costs the host
        # ~1 ms at every size (profiles/r04_c10_graph_host_time.jsonl), with identical results (the call never synchronises or
        # allocates inside the library, so it captures as it is).  One graph per call signature -- device, (B, P, TP), steps, rigidity
        # forcing, arithmetic -- holding static input / output buffers and its own workspace: meant for serving loops whose batches
        # repeat a geometry (the segment tables and work lists are rebuilt on the device inside the graph, so the part sizes may differ
        # from call to call as long as B, P and the point count do not).  A new signature costs one eager call plus the capture;
        # the `graph_cache` most recent ones are kept.  Results are returned as copies (the static buffers belong to the graph).
        self.graph_replay = (os.environ.get("RAP_GRAPH_REPLAY", "0") == "1") if graph_replay is None else bool(graph_replay)
        self.graph_cache = 4
        self._graphs: dict = {}

    # ---- the nn.Module surface the reference's checkpoint loader relies on (sample.py:57-59 -> utils/checkpoint.py:13-61:
    # `load_checkpoint_for_module(model, ckpt_path)` reads ckpt["state_dict"] -- LightningModule keys, i.e. `flow_model.<name>` for
    # the velocity network -- and calls `model.load_state_dict(state_dict, strict=False)`; then `model.eval()`) -------------------
    _PREFIX = "flow_model."

    def state_dict(self) -> dict:
        return {self._PREFIX + k: v for k, v in self.flow_model.state_dict().items()}

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        """LightningModule-keyed state dict -> the velocity network.  Keys under ``flow_model.`` go to ``PointCloudDiT``; anything
        else (``feature_extractor.*`` of a checkpoint trained with the encoder, loss buffers) is reported as unexpected -- an error
        with ``strict=True``, ignored with ``strict=False`` (what the reference's loader passes).  A bare ``{"state_dict": {...}}``
        checkpoint dict is unwrapped.  Returns ``(missing_keys, unexpected_keys)`` like ``nn.Module.load_state_dict``."""
        from .flow_model import _IncompatibleKeys
        if "state_dict" in state_dict and isinstance(state_dict["state_dict"], dict):
            state_dict = state_dict["state_dict"]
        own = {k[len(self._PREFIX):]: v for k, v in state_dict.items() if k.startswith(self._PREFIX)}
        foreign = [k for k in state_dict if not k.startswith(self._PREFIX)]
        res = self.flow_model.load_state_dict(own, strict=False)
        missing = [self._PREFIX + k for k in res.missing_keys]
        unexpected = foreign + [self._PREFIX + k for k in res.unexpected_keys]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for RectifiedPointFlow: missing {missing[:4]}..., "
                               f"unexpected {unexpected[:4]}...")
        return _IncompatibleKeys(missing, unexpected)

    def eval(self):
        return self

    def to(self, device):
        self.flow_model.to(device)
        return self

    def cuda(self, device=None):
        self.flow_model.cuda(device)
        return self

    # modeling.py:203-231 without the boolean-mask compaction (which syncs): empty parts stay in the table as
    # zero-length segments, which every kernel treats as a no-op and which yields the same zero R,t rows.
    @staticmethod
    def _prepare_data(data_dict: dict, latent_features: torch.Tensor | None = None):
        cond = data_dict["pointclouds"]
        _require_cuda(cond, 'data_dict["pointclouds"]')
        device = cond.device
        return dict(
            # latent point features (TP, in_dim) of a flow model built with in_dim > 0 (modeling.py:636, embedding.py:163-166), else None
            latent=None if latent_features is None else _f32c(latent_features.to(device).reshape(cond.shape[0], -1)),
            cond=_f32c(cond), feats=_f32c(data_dict["features"].to(device)), scales=_f32c(data_dict["scales"].to(device)),
            anchor=data_dict["anchor_indices"].to(device=device, dtype=torch.uint8).contiguous(),
            ppp=data_dict["points_per_part"].to(device=device, dtype=torch.int64).contiguous(),
            cu_batch=data_dict["cu_seqlens"].to(device=device, dtype=torch.int32).contiguous())

    def _resolved_streams(self) -> int:
        n = self.num_streams
        if n is None:
            n = 1
        return max(1, int(n))

    @torch.inference_mode()
    def sample_and_register(self, data_dict: dict, x_1: torch.Tensor | None = None,
                            return_transformer_features: bool = False, latent_features: torch.Tensor | None = None) -> dict:
        """One generation: {'end_point_trajectory','trajectory' (S,TP,3), 'R' (B,P,3,3), 't' (B,P,3)[, features]}.

        With ``num_streams`` > 1 the batch is cut at sample boundaries into that many shards of about equal token count, which run
        concurrently on the current stream and on auxiliary streams (forked from / joined back into the current stream with
        events: the call stays stream-ordered for the caller).  The cut needs the token offsets of the samples on the HOST: free
        when ``cu_seqlens`` is a CPU tensor (as the reference's collate delivers it), one (B+1)-int read otherwise."""
        d = self._prepare_data(data_dict, latent_features)
        in_dim = int(getattr(self.flow_model, "in_dim", 0))
        if (d["latent"] is None) != (in_dim == 0) or (d["latent"] is not None and d["latent"].shape[1] != in_dim):
            raise ValueError(f"latent_features must be (TP, {in_dim}) for this flow model" if in_dim else "latent_features must be None (in_dim == 0)")
        B = d["ppp"].shape[0]
        n = min(self._resolved_streams(), B)
        d["flag"] = self._validate(d)             # once for the whole batch, before any shard is forked
        if self.graph_replay and not torch.cuda.is_current_stream_capturing():
            return self._sample_graph(d, x_1, return_transformer_features)
        if n <= 1:
            return self._sample_shard(d, x_1, return_transformer_features)
        cond = d["cond"]
        device = cond.device
        TP = cond.shape[0]
        x_1 = torch.randn_like(cond) if x_1 is None else _f32c(x_1.to(device))
        cu_src = data_dict["cu_seqlens"]
        cu_host = (cu_src if not cu_src.is_cuda else cu_src.cpu()).to(torch.int64).tolist()
        cuts = shard_cuts(cu_host, n)
        if len(cuts) <= 2:
            return self._sample_shard(d, x_1, return_transformer_features)
        cur = torch.cuda.current_stream(device)
        nsh = len(cuts) - 1
        parts = [None] * nsh
        sequential = getattr(self, "_sequential_shards", False)      # experiment knob: the same shards one after the other on one stream
        fork = torch.cuda.Event()
        fork.record(cur)                                             # the inputs are ready on the caller's stream at this point
        # auxiliary shards first, shard 0 last on the caller's stream: nothing enqueued on `cur` after the fork event delays them
        for k in list(range(1, nsh)) + [0]:
            b0, b1 = cuts[k], cuts[k + 1]
            t0, t1 = cu_host[b0], cu_host[b1]
            shard = dict(cond=cond[t0:t1], feats=d["feats"][t0:t1], scales=d["scales"][b0:b1], anchor=d["anchor"][t0:t1],
                         ppp=d["ppp"][b0:b1], cu_batch=None, flag=d["flag"], latent=None if d["latent"] is None else d["latent"][t0:t1])
            if k == 0 or sequential:
                shard["cu_batch"] = d["cu_batch"][: b1 + 1] if k == 0 else (d["cu_batch"][b0: b1 + 1] - int(t0)).contiguous()
                parts[k] = self._sample_shard(shard, x_1[t0:t1], return_transformer_features)
                continue
            key = (device.index, k)
            st = self._aux_streams.get(key)
            if st is None:
                st = self._aux_streams[key] = torch.cuda.Stream(device)
            st.wait_event(fork)
            with torch.cuda.stream(st):
                shard["cu_batch"] = (d["cu_batch"][b0: b1 + 1] - int(t0)).contiguous()
                parts[k] = self._sample_shard(shard, x_1[t0:t1], return_transformer_features)
            for v in (cond, d["feats"], d["scales"], d["anchor"], d["ppp"], d["cu_batch"], x_1) + ((d["flag"],) if d["flag"] is not None else ()) + (
                    (d["latent"],) if d["latent"] is not None else ()):
                v.record_stream(st)                                  # caching-allocator safety: these are read on `st`
        if not sequential:
            for k in range(1, nsh):
                cur.wait_stream(self._aux_streams[(device.index, k)])    # join
        res = {"end_point_trajectory": torch.cat([o["end_point_trajectory"] for o in parts], dim=1),
               "trajectory": torch.cat([o["trajectory"] for o in parts], dim=1),
               "R": torch.cat([o["R"] for o in parts], dim=0), "t": torch.cat([o["t"] for o in parts], dim=0)}
        if return_transformer_features:
            res["transformer_features"] = torch.cat([o["transformer_features"] for o in parts], dim=0)
        for o in parts[1:]:
            for v in o.values():
                v.record_stream(cur)                                 # allocated on an auxiliary stream, last read (the cat) on `cur`
        return res

    # ---- input validation (split_parts' asserts, utils/point_clouds.py:33-52) ------------------------------------------------
    def _validate(self, d: dict):
        """Enqueue rap_check_batch for a prepared batch.  Returns the device flag (deferred mode: the sampling call poisons its
        results with it) or None (eager mode / off / graph capture).  Raises for verdicts of EARLIER calls that have arrived."""
        self.check_pending(block=False)
        if not self.validate_inputs or torch.cuda.is_current_stream_capturing():
            return None
        cond = d["cond"]
        device = cond.device
        B, P = d["ppp"].shape
        lib = _lib.load()
        flag = torch.zeros(1, dtype=torch.int32, device=device)
        _lib.check(lib.rap_check_batch(_lib.ptr(d["ppp"]), _lib.ptr(d["cu_batch"]), B, P, cond.shape[0], _lib.ptr(flag),
                                       _lib.current_stream(device)), "rap_check_batch")
        if self.validate_inputs == "eager":
            bits = int(flag.item())
            if bits:
                raise ValueError(self._inconsistent(bits))
            return None
        if len(self._pending) >= 64:               # a caller that never lets a verdict arrive: settle the oldest ones now
            self.check_pending(block=True)
        if self._pin is None:
            self._pin = torch.zeros(128, dtype=torch.int32).pin_memory()     # ring of verdict slots (> the 64 that can be pending)
        host = self._pin[self._pin_next:self._pin_next + 1]
        self._pin_next = (self._pin_next + 1) % self._pin.numel()
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))       # BEFORE the sampling call is enqueued: ready within microseconds
        self._pending.append((ev, host, f"batch of {B} sample(s), {cond.shape[0]} points"))
        return flag

    @staticmethod
    def _inconsistent(bits: int) -> str:
        return (f"inconsistent batch (flags {bits:#x}): sum(points_per_part) must equal the number of points and "
                "match cu_seqlens per sample (reference: split_parts, utils/point_clouds.py:33-52)")

    def check_pending(self, block: bool = True) -> None:
        """Raise ValueError if a DEFERRED input check of an earlier call failed (its results were overwritten with NaN on the
        device).  ``block=False`` looks only at verdicts that have already arrived (never waits for the GPU)."""
        keep, bad = [], None
        for ev, host, what in self._pending:
            if not block and not ev.query():
                keep.append((ev, host, what))
                continue
            if block:
                ev.synchronize()
            bits = int(host.item())
            if bits and bad is None:
                bad = (bits, what)
        self._pending = keep
        if bad is not None:
            raise ValueError(self._inconsistent(bad[0]) + f" -- reported for an earlier call ({bad[1]}); its results are NaN")

    def synchronize(self) -> None:
        """Wait for the device and surface deferred validation errors (the natural place for a caller that is about to read results)."""
        torch.cuda.synchronize()
        self.check_pending(block=True)

    def _sample_graph(self, d: dict, x_1: torch.Tensor | None, return_transformer_features: bool) -> dict:
        """One sampling call as a replay of a captured HIP graph (see ``graph_replay``).  Validation stays outside the graph (the
        flag kernel before, the poison kernels after: a capture cannot carry the deferred read-back)."""
        cond = d["cond"]
        device = cond.device
        TP = cond.shape[0]
        B, P = d["ppp"].shape
        S = int(self.inference_sampling_steps)
        model = self.flow_model
        model._activate(device)
        lib = _lib.load()
        key = (device.index, B, P, TP, S, bool(self.rigidity_forcing), lib.rap_model_compute_dtype(model._handle),
               lib.rap_model_residual_dtype(model._handle), bool(return_transformer_features), id(model), model._generation)
        x_1 = torch.randn_like(cond) if x_1 is None else _f32c(x_1.to(device))    # modeling.py:664 (outside the graph: fresh noise per call)
        entry = self._graphs.get(key)
        if entry is None:
            static = {k: d[k].clone() for k in ("cond", "feats", "scales", "anchor", "ppp", "cu_batch")}
            static["flag"] = None
            static["latent"] = None if d.get("latent") is None else d["latent"].clone()
            sx = x_1.clone()
            self._sample_shard(static, sx, return_transformer_features)      # eager warm-up: weight copies, attributes, allocator
            torch.cuda.current_stream(device).synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._sample_shard(static, sx, return_transformer_features)
            entry = (graph, static, sx, out)
            while len(self._graphs) >= max(1, int(self.graph_cache)):
                self._graphs.pop(next(iter(self._graphs)))                    # oldest signature first
            self._graphs[key] = entry
        else:
            self._graphs[key] = self._graphs.pop(key)                         # most recently used last
        graph, static, sx, out = entry
        for k in ("cond", "feats", "scales", "anchor", "ppp", "cu_batch"):
            static[k].copy_(d[k])
        if static["latent"] is not None:
            static["latent"].copy_(d["latent"])
        sx.copy_(x_1)
        graph.replay()
        res = {k: v.clone() for k, v in out.items()}
        flag = d.get("flag")
        if flag is not None:
            stream = _lib.current_stream(device)
            with torch.cuda.device(device):
                # every result of an inconsistent batch: poses, ALL trajectory steps, the captured features
                for buf in [res["R"], res["t"], res["end_point_trajectory"], res["trajectory"]] + (
                        [res["transformer_features"]] if "transformer_features" in res else []):
                    _lib.check(lib.rap_poison_on_flag(_lib.ptr(flag), _lib.ptr(buf), buf.numel(), stream), "rap_poison_on_flag")
        return res

    def _sample_shard(self, d: dict, x_1: torch.Tensor | None, return_transformer_features: bool) -> dict:
        """rap_sample on the current stream for one (shard of a) prepared batch."""
        cond = d["cond"]
        device = cond.device
        TP = cond.shape[0]
        B, P = d["ppp"].shape
        S = int(self.inference_sampling_steps)
        x_1 = torch.randn_like(cond) if x_1 is None else _f32c(x_1.to(device))    # modeling.py:664
        model = self.flow_model
        model._activate(device)
        lib = _lib.load()
        traj_x0 = torch.empty((S, TP, 3), dtype=torch.float32, device=device)     # sampler.py:47-49
        traj_xt = torch.empty((S, TP, 3), dtype=torch.float32, device=device)
        R = torch.empty((B, P, 3, 3), dtype=torch.float32, device=device)
        t = torch.empty((B, P, 3), dtype=torch.float32, device=device)
        feats_out = torch.empty((TP, model.embed_dim), dtype=torch.float32, device=device) if return_transformer_features else None
        ws = workspace(device, lib.rap_workspace_bytes(model._handle, TP, B, B * P, S))
        stream = _lib.current_stream(device)
        with torch.cuda.device(device):
            feats = d["feats"] if getattr(model, "_native_feat", 1) else None      # local_feat_concat_on=False: the features are not an input
            rc = lib.rap_sample_latent(model._handle, _lib.ptr(cond), _lib.ptr(feats), _lib.ptr(d.get("latent")), _lib.ptr(d["scales"]),
                                       _lib.ptr(d["anchor"]), _lib.ptr(d["ppp"]), _lib.ptr(d["cu_batch"]), _lib.ptr(x_1), B, P, TP, S,
                                       1 if self.rigidity_forcing else 0, _lib.ptr(traj_x0), _lib.ptr(traj_xt), _lib.ptr(R),
                                       _lib.ptr(t), _lib.ptr(feats_out), _lib.ptr(ws), ws.numel(), stream)
            _lib.check(rc, "rap_sample")
            flag = d.get("flag")
            if flag is not None:
                # deferred validation: an inconsistent batch yields NaN poses, NaN clouds at EVERY flow step and NaN features -- never
                # plausible numbers (round 4 poisoned only the poses and the last step)
                for buf in [R, t, traj_x0, traj_xt] + ([feats_out] if feats_out is not None else []):
                    _lib.check(lib.rap_poison_on_flag(_lib.ptr(flag), _lib.ptr(buf), buf.numel(), stream), "rap_poison_on_flag")
        out = {"end_point_trajectory": traj_x0, "trajectory": traj_xt, "R": R, "t": t}
        if return_transformer_features:
            out["transformer_features"] = feats_out
        return out

    @torch.inference_mode()
    def sample_rectified_flow(self, data_dict: dict, latent_features: torch.Tensor | None, x_1: torch.Tensor | None = None,
                              return_tarjectory: bool = False, return_transformer_features: bool = False):
        """Reference signature and return structure (modeling.py:633-741; the misspelt keyword is the reference's).

        The per-part poses of the final end point -- what ``test_step`` computes next with
        ``fit_transformations(cond, trajs[-1], ...)`` (modeling.py:389-391) -- come out of the same call and are
        kept in ``self.last_poses`` as ``(R (B,P,3,3), t (B,P,3))``."""
        out = self.sample_and_register(data_dict, x_1, return_transformer_features, latent_features=latent_features)
        self.last_poses = (out["R"], out["t"])
        result = {"end_point_trajectory": out["end_point_trajectory"], "trajectory": out["trajectory"]}
        if return_transformer_features:
            if return_tarjectory:
                return {"trajectory": result, "transformer_features": out["transformer_features"]}
            # modeling.py:733 indexes the dict with [-1] (a latent bug: KeyError); we return the final end point
            return {"points": result["end_point_trajectory"][-1], "transformer_features": out["transformer_features"]}
        return result

    @torch.inference_mode()
    def sample_generations(self, data_dict: dict, x_1_list=None, use_average_rigidity_rmse: bool = True,
                           batch_generations: bool | None = None, latent_features: torch.Tensor | None = None) -> dict:
        """``n_generations`` samples per object + the reference's rigidity-based selection (test_step, modeling.py:397-592):
        per generation the trajectory, final cloud and poses; per object the generation whose rigidity RMSE is smallest --
        averaged over all end-point trajectory steps (``use_average_rigidity_rmse``, modeling.py:466-500) or taken at the
        final step (:501-504).  Everything stays on the device.

        The reference loops the generations one after the other (modeling.py:351-361).  Generations are independent samples of
        the same objects (same condition / features, their own x_1), so here the G x B samples ride in ONE ``rap_sample`` call
        (round 4; ``batch_generations=False`` or more than 65 535 parts restore the loop): with the shipped ``batch_size: 1`` a
        single generation leaves the GPU at the launch floor, and G of them cost about what one does.  Each generation's noise is
        drawn exactly as the loop would draw it (one ``randn_like`` per generation, in order).  The rigidity RMSEs of all
        generations come out of one trajectory pass over the stacked batch."""
        from .selection import (average_trajectory_rigidity_rmse, compute_rigidity_rmse, select_generations_by_rigidity)
        G = int(self.n_generations)
        d = self._prepare_data(data_dict, latent_features)
        cond, ppp, cu, scales = d["cond"], d["ppp"], d["cu_batch"], d["scales"]
        B, P = ppp.shape
        TP = cond.shape[0]
        if batch_generations is None:
            batch_generations = os.environ.get("RAP_BATCH_GENERATIONS", "1") != "0"
        stacked_call = batch_generations and G > 1 and G * B * P <= 65535 and G * TP <= 0x7fffffff // 8
        if stacked_call:
            # memory gate: the stacked call needs the workspace AND the trajectories of G x TP points at once (fp32: ~23 KB of
            # workspace per point + 24 B per point per flow step); near-limit batches with several generations would exhaust the device where
            # the loop fits.  Stack only when that fits into half of the memory that is free right now.
            model = self.flow_model
            model._activate(cond.device)
            S_ = int(self.inference_sampling_steps)
            need = (_lib.load().rap_workspace_bytes(model._handle, G * TP, G * B, G * B * P, S_) + 2 * S_ * G * TP * 12
                    + G * TP * (24 + 4 * (d["feats"].shape[1] if d["feats"].dim() == 2 else 0)))
            free, _total = torch.cuda.mem_get_info(cond.device)
            stacked_call = need <= (free + torch.cuda.memory_reserved(cond.device) - torch.cuda.memory_allocated(cond.device)) // 2
        avg = use_average_rigidity_rmse and self.return_end_point_trajectory
        if stacked_call:
            device = cond.device
            x_1 = torch.cat([torch.randn_like(cond) if x_1_list is None or x_1_list[g] is None else _f32c(x_1_list[g].to(device))
                             for g in range(G)])
            offs = (torch.arange(G, device=device, dtype=torch.int32) * TP)[:, None]
            big = dict(cond=cond.repeat(G, 1), feats=d["feats"].repeat(G, 1), scales=scales.repeat(G), anchor=d["anchor"].repeat(G),
                       ppp=ppp.repeat(G, 1), latent=None if d["latent"] is None else d["latent"].repeat(G, 1),
                       cu_batch=torch.cat([(cu[:-1][None, :] + offs).reshape(-1), cu.new_full((1,), G * TP)]).contiguous(),
                       flag=self._validate(d))                       # the G copies are consistent iff the batch is
            o = self._sample_shard(big, x_1, False)   # one stream: the stacked call fills the chip; shards would only cut it up again
            ep, tr = o["end_point_trajectory"], o["trajectory"]
            Rg, tg = o["R"].view(G, B, P, 3, 3), o["t"].view(G, B, P, 3)
            gens = [{"end_point_trajectory": ep[:, g * TP:(g + 1) * TP], "trajectory": tr[:, g * TP:(g + 1) * TP],
                     "R": Rg[g], "t": tg[g]} for g in range(G)]
            if avg:
                stacked = average_trajectory_rigidity_rmse(big["cond"], ep, big["ppp"], big["cu_batch"], big["scales"]).view(G, B)
            else:
                stacked = compute_rigidity_rmse(big["cond"], ep[-1], o["R"], o["t"], big["ppp"], big["cu_batch"], big["scales"]).view(G, B)
            finals = ep[-1].view(G, TP, 3)
        else:
            gens = []
            for g in range(G):
                x_1 = None if x_1_list is None else x_1_list[g]
                gens.append(self.sample_and_register(data_dict, x_1=x_1, latent_features=latent_features))
            if avg:
                rig = [average_trajectory_rigidity_rmse(cond, o["end_point_trajectory"], ppp, cu, scales) for o in gens]
            else:
                rig = [compute_rigidity_rmse(cond, o["end_point_trajectory"][-1], o["R"], o["t"], ppp, cu, scales) for o in gens]
            stacked = torch.stack(rig)                                                          # (G,B)
            finals = torch.stack([o["end_point_trajectory"][-1] for o in gens])
            Rg, tg = torch.stack([o["R"] for o in gens]), torch.stack([o["t"] for o in gens])
        best, cloud, R, t = select_generations_by_rigidity(stacked, finals, Rg, tg, cu)
        return {"generations": gens, "rigidity_rmse": stacked, "best_gen_indices": best, "pointclouds_selected": cloud,
                "rotations_selected": R, "translations_selected": t, "generations_in_one_call": bool(stacked_call)}

    sample = sample_rectified_flow   # the name BASELINE.json's north_star uses
