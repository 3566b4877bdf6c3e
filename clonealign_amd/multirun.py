"""Multi-restart dispatch for ``run_clonealign`` (R/clonealign.R:50-56).

The reference runs ``length(initial_shrinks) * n_repeats`` fits sequentially.  Restarts
are independent ("replicas only"): here they are dealt over the
visible GPUs, one fit per GPU at a time, each on its own engine handle and HIP stream.
No collective is involved; the host picks ``which.max(final_elbo)`` afterwards.

The restarts of one GPU share ONE engine: the first uploads the count matrix and computes its fit constants, the later
ones restart it (``ca_reinit``: initial values, fresh Adam state) and skip the host passes over the matrix -- what the
reference repeats nine times by calling ``clonealign()`` again is done once per GPU.
"""
from concurrent.futures import ThreadPoolExecutor


def run_restarts(gene_expression_data, copy_number_data, jobs, seeds, devices, kwargs):
    from .api import clonealign
    devices = list(devices) if devices else [None]
    base_opts = dict(kwargs.pop("engine_opts", None) or {})
    # per GPU: prepared inputs + the resident engine, shared by that GPU's restarts (only with the default HIP engine)
    shared = [({} if kwargs.get("engine") is None else None) for _ in devices]

    def one(i):
        kw = dict(kwargs)
        kw.update(jobs[i])
        opts = dict(base_opts)
        dev = devices[i % len(devices)]
        if dev is not None:
            opts["device"] = int(dev)
        return clonealign(gene_expression_data, copy_number_data, seed=seeds[i],
                          engine_opts=opts or None, _reuse=shared[i % len(devices)], **kw)

    try:
        if len(devices) == 1:
            return [one(i) for i in range(len(jobs))]
        # one worker thread per GPU; job i runs on devices[i % D] so each GPU has one fit in flight
        fits = [None] * len(jobs)

        def lane(d):
            for i in range(d, len(jobs), len(devices)):
                fits[i] = one(i)

        with ThreadPoolExecutor(max_workers=len(devices)) as ex:
            list(ex.map(lane, range(len(devices))))
        return fits
    finally:
        for sh in shared:
            eng = (sh or {}).pop("eng", None)
            if eng is not None:
                eng.close()
