"""Data parallelism of the RAT models: what one training step puts on the links and how — the dense-net all-reduce, the
table gradients as row lists (owner-partitioned all-to-all + all-gather, or all-gather at capacity + merge), SyncBN's statistics and the
small helpers every collective goes through (`_collective`: under a captured step a collective is an eager closure between two graph
segments).  A mixin of rat_amd.model.RAT_m2 (split out of model.py in round 6; no behaviour change): it reads the model's flat buffers,
sort plans and configuration through `self`.  RCCL on device tensors (backend "nccl"), gloo staged through the host (CPU tests)."""
import torch

from . import ops


class DataParallelExchange:
    def _exchange_gradients(self, g=None):
        """Data parallelism.  C1: all-reduce (RCCL over xGMI) of the dense-net slice of the flat gradient bucket —
        started inside backward, only waited for here.  Tables: in sparse mode, and in the dense modes whenever the fused training
        step asked backward for row lists (`_table_lists`), C2: all-gather of (row ids, gradient rows, count) + the same
        deterministic plan + segmented reduction on every rank (28.8 MB per rank at the north-star strong-scaling shape instead of
        a 257 MB all-reduce; the regulariser's lambda*W is replica-identical and never travels); otherwise one dense all-reduce.
        `g`: the flat gradient buffer when the caller holds it (the fused step; p.grad is not populated there)."""
        if not self._dp():
            return
        explicit = g is not None
        if g is None:
            g = self._gather_flat_grad()
        pending, self._pending_reduce = self._pending_reduce, None
        lists, self._table_lists = self._table_lists, None
        world = self._world_size()
        ring = lambda nbytes: int(2 * nbytes * (world - 1) / world)      # noqa: E731  (an all-reduce moves ~2 (N-1)/N of its size per rank)
        if g is not None:
            n0 = self._n_emb - self._n_sparse                      # [0, n0): "embedding_layer" tensors with a dense gradient
            form = ("owner_lists" if self._owner_state is not None else "gathered_lists") if lists is not None else \
                ("owner_lists(sparse)" if self._owner_state is not None else "gathered_lists(sparse)") if self._sparse is not None else \
                "dense_allreduce"
            self._exchange_info = dict(form=form, world=world, dense_net_bytes=ring(4 * (g.numel() - n0)),
                                       table_bytes=ring(4 * n0) if form == "dense_allreduce" else None)
            if pending is not None and pending[1] is g:            # the dense-net part is already on its way
                if lists is not None:
                    if self._owner_state is not None:              # (the label table's partial gradients ride in the lists' headers)
                        self._exchange_lists_owner(g, lists)
                    else:
                        for part in lists:                         # into the zeroed table block
                            rows, grads, count, width, _t, _b = self._merge_sparse(part)
                            ops.scatter_rows(g[part[5]:], rows, grads, count, width, lib=self._lib)
                        if n0 > self._n_tab:
                            self._all_reduce_sum(g[self._n_tab:n0])    # the label table (3 x d floats)
                elif n0 > 0:
                    self._all_reduce_sum(g[:n0])
                self._collective(lambda: pending[0][0].wait())
            else:
                if pending is not None:
                    pending[0][0].wait()
                    raise RuntimeError("gradient buffers were replaced between backward and the exchange: the in-flight all-reduce "
                                       "of the dense-net gradients would be summed twice (gradient accumulation under data "
                                       "parallelism is not supported on this path)")
                assert lists is None
                self._all_reduce_sum(g)
            if not explicit and g is not self._last_gflat:
                for n in self._dense_names():
                    self._params[n].grad = self._gflat_view(g, n)
                self._last_gflat = g
        if self._sparse is not None and not self._sparse_is_global:      # (a second call must not merge global lists again)
            if self._owner_state is not None and g is not None:
                self._sparse = self._exchange_lists_owner(g, self._sparse)
            else:
                self._sparse = [self._merge_sparse(part) for part in self._sparse]
            self._sparse_is_global = True
        self._owner_state = None
        info = self.__dict__.get("_exchange_info")
        if info is not None and info["table_bytes"] is None:
            if info["form"].startswith("owner"):
                info["table_bytes"] = (self.__dict__.get("_owner_stats") or {}).get("wire_bytes")
            elif lists is not None:                                # all-gather at capacity: my lists to every peer
                info["table_bytes"] = sum(4 * (p[0].numel() * (1 + p[3]) + 1) for p in lists) * (world - 1)

    # Owner-partitioned exchange of the table-gradient lists: rank k owns the rows [k R/N, (k+1) R/N) of a table family; every rank sends
    # each owner ITS rows of the local (sorted, unique) lists — an all-to-all of exactly the pairs that exist — the owner sorts / reduces
    # only what it received (1/N of the union instead of the whole union on every rank) and the reduced lists are all-gathered.  Same
    # sums in the same (rank) order as the all-gather form, so the replicas stay bit-identical.
    #
    # Round 5: no host stall, three collectives per step instead of twelve.  The split sizes of the all-to-all are the per-owner counts
    # of the local lists — a function of the batch's IDS alone.  `_owner_prepare` therefore builds the sort plans and the counts at
    # the START of the step (rat_owner_counts), all-gathers the N x N matrix and starts its copy to pinned host memory; the host reads
    # it when the backward has been enqueued — long after the copy has landed.  Both table families and the label table's partial
    # gradient share ONE wire buffer per peer (rat_owner_pack / rat_owner_unpack) and ONE all-gathered list per rank
    # (rat_owner_scatter).  Under a captured step the exchange is one eager closure between two graph segments (its buffer sizes
    # change from step to step; its inputs and outputs — the local lists, the gradient bucket — are static).
    owner_exchange = True
    _OWNER_BUCKET = 4096                 # list capacities are rounded up to this many rows (few distinct plan sizes)

    def _merge_sparse(self, part):
        """one family's local (rows, grads, count) -> the global list as ONE (rows, grads, count, width, total_rows, base) record
        (all-gather at capacity + merge of the union on every rank: the form without owners)"""
        return self._merge_sparse_gather(part)

    def _use_owner_exchange(self):
        # (sparse mode inside a captured step: the optimizer's row kernels are recorded with the list pointers baked in — the
        # all-gather form has static capacities)
        return bool(self.owner_exchange) and not (self._tape is not None and self._grad_mode == "sparse")

    def _raw_all_to_all(self, out, inp, out_splits, in_splits):
        """uneven all-to-all along dim 0 (RCCL on device tensors; gloo on host tensors, staged through the host for device tensors)"""
        import torch.distributed as dist
        if self._staged(inp):
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(ho, inp.detach().cpu().contiguous(), out_splits, in_splits)
            out.copy_(ho)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits)

    def _raw_all_gather(self, out, t):
        import torch.distributed as dist
        world = self._world_size()
        if self._staged(t):
            parts = [torch.empty(t.numel(), dtype=t.dtype) for _ in range(world)]
            dist.all_gather(parts, t.detach().reshape(-1).cpu())
            out.copy_(torch.cat(parts))
        elif t.is_cuda:
            dist.all_gather_into_tensor(out, t)
        else:
            dist.all_gather(list(out.view(world, -1).unbind(0)), t)

    def _build_plans(self, idx, dims):
        """the sort plans of a batch's table rows (feature tables; LR tables): functions of the ids alone"""
        c, lib = self._cfg, self._lib
        B, T, L, S = dims
        d, F = c["d"], c["nf"]
        if self._col2field is None or self._col2field.numel() != L:
            self._col2field = ops.col2field_table(self._fields, L, idx.device)
        rows_feat = self._n_feat // d
        # one plan per batch size, kept for the model's life: captured steps point into its workspace (see _merge_sparse_gather)
        plan = ops.sparse_plan_ids(idx, self._ftab, self._col2field, F, self._flat, d, rows_feat, B, T, L,
                                   plan=self._ws.get(("plan", 0, B * T * L)), lib=lib)
        self._ws[("plan", 0, B * T * L)] = plan
        plan_lr = None
        if c["use_wide"]:
            rows_lr = self._n_tab - self._n_feat
            plan_lr = ops.sparse_plan_ids(idx, self._lr_ftab, self._col2field, F, self._flat[self._n_feat:], 1, rows_lr, B, T, L,
                                          target_only=True, plan=self._ws.get(("plan", 1, B * L)), lib=lib)
            self._ws[("plan", 1, B * L)] = plan_lr
        return plan, plan_lr

    def _owner_prepare(self, idx, dims):
        """Start of a fused iteration whose table gradients will travel as row lists: plans, per-owner counts, the N x N count matrix
        on every rank and — asynchronously — on every host."""
        c, lib = self._cfg, self._lib
        world, dev = self._world_size(), idx.device
        d = c["d"]
        plans = self._build_plans(idx, dims)
        per_a = -(-(self._n_feat // d) // world)
        per_b = -(-(self._n_tab - self._n_feat) // world) if plans[1] is not None else 1
        cnt = self._ws.get(("owner-counts", world))
        if cnt is None or cnt.device != dev:
            cnt = self._ws[("owner-counts", world)] = torch.zeros(2 * world, dtype=torch.int32, device=dev)
        ops.owner_counts(plans[0], per_a, world, cnt[:world], lib=lib)
        if plans[1] is not None:
            ops.owner_counts(plans[1], per_b, world, cnt[world:], lib=lib)
        self._owner_publish(cnt, plans, (per_a, per_b))

    def _owner_publish(self, cnt, plans, per):
        """cnt: int32 [2][world] on the device, this rank's pairs per (family, owner) -> `_owner_state`"""
        import torch.distributed as dist
        world, rank, dev = self._world_size(), dist.get_rank(), cnt.device
        mat = torch.empty(2 * world * world, dtype=torch.int32, device=dev)
        host = self._ws.get(("owner-host", world))
        if host is None:
            host = torch.empty(2 * world * world, dtype=torch.int32)
            host = self._ws[("owner-host", world)] = host.pin_memory() if dev.type == "cuda" else host
        event = torch.cuda.Event() if dev.type == "cuda" else None

        def run():
            self._raw_all_gather(mat, cnt)
            host.copy_(mat, non_blocking=True)
            if event is not None:
                event.record()
        self._collective(run)
        self._owner_state = dict(plans=plans, mat=mat, host=host, event=event, per=per, rank=rank, world=world)

    def _exchange_lists_owner(self, g, lists):
        """lists: this backward's [(rows, grads, count, width, total_rows, base)] per table family (feature tables[, LR tables]) ->
        dense modes: the global gradient rows land in the zeroed table block of `g` and the label table's gradient is summed over
        the ranks; sparse mode: returns the global lists as records for the row optimizer."""
        st, self._owner_state = self._owner_state, None
        c, lib = self._cfg, self._lib
        d, world, rank = c["d"], st["world"], st["rank"]
        mat, host, event = st["mat"], st["host"], st["event"]
        rows_a, grads_a, _ca, _wa, total_a, base_a = lists[0]
        rows_b, vals_b, _cb, _wb, total_b, base_b = lists[1] if len(lists) > 1 else (None, None, None, 1, 0, 0)
        dev = rows_a.device
        sparse_mode = self._grad_mode == "sparse"
        # the label table's slot (the "embedding_layer" tensor without a row list): its partial gradients ride in the lists' headers
        # (sparse mode: it is part of the dense slice _exchange_gradients all-reduces)
        n_label = 0 if sparse_mode else self._n_emb - self._n_tab
        label = g[self._n_tab:self._n_emb] if n_label > 0 else None
        pad4 = lambda n: (n + 3) // 4 * 4                              # noqa: E731
        chunk = lambda na, nb: pad4(na) + na * d + 2 * pad4(nb)        # noqa: E731
        bucket = lambda n: max(self._OWNER_BUCKET, -(-n // self._OWNER_BUCKET) * self._OWNER_BUCKET)      # noqa: E731
        out = {}

        def run():
            recording = self._tape is not None        # graph capture: nothing has executed, the matrix is not there yet — issue the
            if recording:                              # same collectives on token buffers and launch nothing that reads it
                S = torch.zeros((world, 2, world), dtype=torch.int64)
            else:
                if event is not None:
                    event.synchronize()
                S = host.view(world, 2, world).to(torch.int64)
            in_splits = [chunk(int(S[rank, 0, k]), int(S[rank, 1, k])) if not recording else 4 for k in range(world)]
            out_splits = [chunk(int(S[k, 0, rank]), int(S[k, 1, rank])) if not recording else 4 for k in range(world)]
            n_send, n_recv = sum(in_splits), sum(out_splits)
            cap_a = bucket(int(S[:, 0, :].sum(0).max()))
            cap_b = bucket(int(S[:, 1, :].sum(0).max())) if rows_b is not None else 0
            max_pairs = int(S.sum(1).max())
            send = torch.empty(max(n_send, 4), dtype=torch.float32, device=dev)
            recv = torch.empty(max(n_recv, 4), dtype=torch.float32, device=dev)
            if not recording:
                ops.owner_pack(mat, world, rank, d, rows_a, grads_a, rows_b, vals_b, max_pairs, send, lib=lib)
            self._raw_all_to_all(recv[:n_recv], send[:n_send], out_splits, in_splits)
            stride = 4 + pad4(n_label) + cap_a * (1 + d) + 2 * cap_b
            mine = torch.empty(stride, dtype=torch.float32, device=dev)
            mine_i = mine.view(torch.int32)
            o_ra = 4 + pad4(n_label)
            o_ga, o_rb = o_ra + cap_a, o_ra + cap_a * (1 + d)
            o_vb = o_rb + cap_b
            if not recording:
                got_ra = torch.empty(cap_a, dtype=torch.int32, device=dev)
                got_ga = torch.empty((cap_a, d), dtype=torch.float32, device=dev)
                got_rb = torch.empty(cap_b, dtype=torch.int32, device=dev) if cap_b else None
                got_vb = torch.empty(cap_b, dtype=torch.float32, device=dev) if cap_b else None
                totals = torch.empty(2, dtype=torch.int32, device=dev)
                if not cap_b:
                    mine_i[1:2].zero_()
                ops.owner_unpack(mat, world, rank, d, recv, max_pairs, got_ra, got_ga, got_rb, got_vb, totals,
                                 extra_src=label, extra_dst=mine[4:4 + n_label] if n_label > 0 else None, lib=lib)
                # the owner's merge: sort + fixed-order reduction of what it received (sources in rank order), written straight into
                # the list this rank contributes to the all-gather
                key = ("merge-owner", 0, cap_a)
                plan = ops.sparse_plan_rows(got_ra, totals[0:1], cap_a, 1, total_a, plan=self._ws.get(key), count_out=mine_i[0:1], lib=lib)
                self._ws[key] = plan
                ops.sparse_reduce_rows(plan, got_ga, cap_a, 1, d, mine_i[o_ra:o_ga], mine[o_ga:o_rb].view(cap_a, d), count=mine_i[0:1], lib=lib)
                if cap_b:
                    key = ("merge-owner", 1, cap_b)
                    plan = ops.sparse_plan_rows(got_rb, totals[1:2], cap_b, 1, total_b, plan=self._ws.get(key), count_out=mine_i[1:2], lib=lib)
                    self._ws[key] = plan
                    ops.sparse_reduce_rows(plan, got_vb.view(cap_b, 1), cap_b, 1, 1, mine_i[o_rb:o_vb], mine[o_vb:].view(cap_b, 1),
                                           count=mine_i[1:2], lib=lib)
            everyone = torch.empty(world * stride, dtype=torch.float32, device=dev)
            self._raw_all_gather(everyone, mine)
            self._owner_stats = dict(sent=int(S[rank].sum()), received=int(S[:, :, rank].sum()), capacity=(cap_a, cap_b),
                                     local_capacity=rows_a.numel(), floats_sent=n_send, floats_gathered=world * stride, collectives=3,
                                     # what this rank puts on the links: its chunks for the other owners + its reduced list to every peer
                                     wire_bytes=4 * ((n_send - in_splits[rank]) + stride * (world - 1)))
            if recording:
                return
            if sparse_mode:
                ev_i = everyone.view(torch.int32).view(world, stride)
                ev_f = everyone.view(world, stride)
                recs = []
                for k in range(world):
                    recs.append((ev_i[k, o_ra:o_ga], ev_f[k, o_ga:o_rb].view(cap_a, d), ev_i[k, 0:1], d, total_a, base_a))
                    if cap_b:
                        recs.append((ev_i[k, o_rb:o_vb], ev_f[k, o_vb:].view(cap_b, 1), ev_i[k, 1:2], 1, total_b, base_b))
                out["records"] = recs
            else:
                ops.owner_scatter(g[base_a:], g[base_b:] if cap_b else None, label, everyone, stride, world, cap_a, cap_b, d, n_label, lib=lib)
        self._collective(run)
        return out.get("records")

    def _merge_sparse_gather(self, part):
        """all-gather one family's (rows, grads, count) at capacity and reduce the union: -> the same record, global"""
        lib, world = self._lib, self._world_size()
        rows, grads, count, width, total_rows, base_off = part
        cap = rows.numel()
        all_rows = self._all_gather_flat(rows)
        all_grads = self._all_gather_flat(grads.reshape(-1))
        all_counts = self._all_gather_flat(count)
        # plans are cached PER SIZE and never dropped: a captured step (graph.StepGraph) has the plan's workspace / count
        # pointers baked in, and a second batch shape (an epoch's tail batch) must not hand that memory back to the allocator
        pkey = ("merge", width, cap * world)
        plan = ops.sparse_plan_rows(all_rows, all_counts, cap, world, total_rows, plan=self._ws.get(pkey), lib=lib)
        self._ws[pkey] = plan
        ncap = min(cap * world, total_rows)
        out_rows = torch.empty(ncap, dtype=torch.int32, device=rows.device)
        out_grads = torch.empty((ncap, width), dtype=torch.float32, device=rows.device)
        ops.sparse_reduce_rows(plan, all_grads, cap, world, width, out_rows, out_grads, lib=lib)
        return (out_rows, out_grads, plan.count.clone(), width, total_rows, base_off)

    def _collective(self, fn):
        """Every communication call of the step goes through here as a closure over tensors that already exist.  Normally it just
        runs; while the step is being captured into hipGraphs (graph.StepGraph) the capture is suspended around it and the closure is
        kept, to be run again between the graph segments of every replay."""
        return self._tape.between_segments(fn) if self._tape is not None else fn()

    def _staged(self, t):
        """device tensors under the gloo backend (GPU tests that run two ranks on ONE device, where RCCL refuses): the collective
        goes through host copies.  RCCL ("nccl") and CPU tensors are used directly."""
        import torch.distributed as dist
        return t.is_cuda and dist.get_backend() == "gloo"

    def _all_gather_flat(self, t):
        """[n] -> [world * n], ranks in order (RCCL all-gather on the GPU, gloo in the CPU tests)."""
        import torch.distributed as dist
        world = self._world_size()
        out = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
        if self._staged(t):
            def run():
                parts = [torch.empty(t.numel(), dtype=t.dtype) for _ in range(world)]
                dist.all_gather(parts, t.detach().reshape(-1).cpu())
                out.copy_(torch.cat(parts))
            self._collective(run)
        elif t.is_cuda:
            self._collective(lambda: dist.all_gather_into_tensor(out, t))
        else:
            self._collective(lambda: dist.all_gather(list(out.view(world, -1).unbind(0)), t))
        return out

    def _all_reduce_sum(self, t):
        import torch.distributed as dist
        if self._staged(t):
            def run():
                h = t.detach().cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h)
            self._collective(run)
        else:
            self._collective(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM))
        return t
