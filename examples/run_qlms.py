"""Filtering -> quadratic estimators -> mean fields -> spectra driver, CLI and job structure of the reference's
examples/run_qlms.py (:25-121), one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/run_qlms.py params/idealized_example.py -imin 0 -imax 255 -k p -ivt -ivp -dd

Every phase shards its jobs as jobs[rank::size] and ends on a barrier, exactly like the reference under srun; the
products land in the same cache files, so a killed run is simply re-run.
"""
import argparse
import os
import sys
from importlib.machinery import SourceFileLoader

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plancklens_amd.helpers import mpi  # noqa: E402

parser = argparse.ArgumentParser(description='QE calculation driver (MI355X)')
parser.add_argument('parfile', type=str, nargs=1)
parser.add_argument('-imin', dest='imin', default=-1, type=int, help='starting index (-1 stands for data map)')
parser.add_argument('-imax', dest='imax', default=-2, type=int, help='last index')
parser.add_argument('-k', dest='k', action='store', default=[], nargs='+', help='QE keys (gradient and curl come together)')
parser.add_argument('-kA', dest='kA', action='store', default=[], nargs='+', help='QE spectra keys (left leg)')
parser.add_argument('-kB', dest='kB', action='store', default=[], nargs='+', help='QE spectra keys (right leg)')
parser.add_argument('-ivt', dest='ivt', action='store_true', help='do T. filtering')
parser.add_argument('-ivp', dest='ivp', action='store_true', help='do P. filtering')
parser.add_argument('-dd', dest='dd', action='store_true', help='perform dd qlms / qcls library QEs')
parser.add_argument('-ds', dest='ds', action='store_true', help='perform ds qlms / qcls library QEs')
parser.add_argument('-ss', dest='ss', action='store_true', help='perform ss qlms / qcls library QEs')
parser.add_argument('-mfdd', dest='mfdd', action='store_true', help='perform dd qlms mean-fields for qcls keys')


def main():
    args = parser.parse_args()
    mpi.init()
    par = SourceFileLoader('run_qlms_parfile', args.parfile[0]).load_module()

    # --- filtering
    jobs = []
    for flag, lab in ((args.ivt, 't'), (args.ivp, 'p')):
        if flag:
            jobs += [(idx, lab) for idx in range(args.imin, args.imax + 1)]
            if args.ds and args.imin >= 0:
                jobs += [(-1, lab)]
    if args.ivt and args.ivp and hasattr(par.ivfs, 'filter_sims'):
        # both filters of a simulation on the same rank (the reference shards the (simulation, field) jobs, run_qlms.py:57): the
        # conjugate-gradient library then runs the temperature and polarization solves of a block at the same time on two streams
        sims_all = sorted(set(idx for idx, _ in jobs))
        mine = [(idx, lab) for idx in sims_all[mpi.rank::mpi.size] for lab in ('t', 'p')]
        idxs = sims_all[mpi.rank::mpi.size]
        if idxs and par.ivfs.filter_sims(idxs, fields='tp'):
            print('rank %s filtered sims %s (t and p) in overlapped block solves' % (mpi.rank, idxs))
            mine = []
    else:
        mine = jobs[mpi.rank::mpi.size]
    if hasattr(par.ivfs, 'filter_sims'):  # conjugate-gradient filters: this rank's simulations in block solves (several per solve)
        for lab in ('t', 'p'):
            idxs = [idx for idx, l in mine if l == lab]
            if idxs and par.ivfs.filter_sims(idxs, fields=lab):
                print('rank %s filtered sims %s %s in block solves' % (mpi.rank, idxs, lab))
                mine = [(idx, l) for idx, l in mine if l != lab]
    for i, (idx, lab) in enumerate(mine):
        print('rank %s filtering sim %s %s, job %s in %s' % (mpi.rank, idx, lab, i, len(jobs[mpi.rank::mpi.size])))
        if lab == 't':
            par.ivfs.get_sim_tlm(idx)
        else:
            par.ivfs.get_sim_elm(idx)  # caches blm as well
    mpi.barrier()

    # --- unnormalized QE calculation
    qlibs = [par.qlms_dd] * args.dd + [par.qlms_ss] * args.ss + [par.qlms_ds] * args.ds
    jobs = [(qlib, idx, k) for qlib in qlibs for k in args.k for idx in range(args.imin, args.imax + 1)]
    mine = jobs[mpi.rank::mpi.size]
    # this rank's jobs, library by library and key by key: get_sim_qlms evaluates the simulations that are not cached yet two at
    # a time where the library can (two simulations on one Legendre recursion); results and cache files are the same
    groups = {}
    for qlib, idx, k in mine:
        groups.setdefault((id(qlib), k), (qlib, k, []))[2].append(idx)
    for qlib, k, idxs in groups.values():
        print('rank %s doing QE sims %s %s, qlm_lib %s (%s of %s jobs)' % (mpi.rank, idxs, k, qlib.lib_dir, len(mine), len(jobs)))
        if hasattr(qlib, 'get_sim_qlms'):
            qlib.get_sim_qlms(k, idxs)
        else:
            for idx in idxs:
                qlib.get_sim_qlm(k, idx)
    mpi.barrier()

    # --- mean-fields: every rank takes part in every mean field -- get_sim_qlm_mf(..., collective=True) shards the simulations of
    # one mean field over the ranks and sums with one all-reduce (the reference shards the (key, half) jobs instead,
    # run_qlms.py:92-95, each rank looping over all simulations of its job).  The call is collective: all ranks make it, in the
    # same order, before the rank-sharded spectra loop below, which then finds the cached mean fields.
    if args.mfdd:
        keys = list(np.unique(np.concatenate([args.kA, args.kB])))
        jobs = [(k, 0) for k in keys] + [(k, 1) for k in keys]
        for i, (k, id0) in enumerate(jobs):
            print("rank %s doing its share of %s QE MF %s" % (mpi.rank, k, id0))
            par.qlms_dd.get_sim_qlm_mf(k, par.qcls_dd.mc_sims_mf[id0::2], collective=True)
    mpi.barrier()

    # --- unnormalized QE power spectra
    qlibs = [par.qcls_dd] * args.dd + [par.qcls_ss] * args.ss + [par.qcls_ds] * args.ds
    jobs = [(qlib, idx, kA, kB) for qlib in qlibs for kA in args.kA for kB in args.kB for idx in range(args.imin, args.imax)
            if idx not in qlib.mc_sims_mf]
    for i, (qlib, idx, kA, kB) in enumerate(jobs[mpi.rank::mpi.size]):
        print('rank %s doing QE spectra sim %s %s %s, qcl_lib %s, job %s in %s' % (mpi.rank, idx, kA, kB, qlib.lib_dir, i, len(jobs)))
        qlib.get_sim_qcl(kA, idx, k2=kB)
    mpi.barrier()
    mpi.finalize()


if __name__ == '__main__':
    main()
