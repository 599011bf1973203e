#!/usr/bin/env python
"""
The reference's examples/p53/p53.py on the GPU: read the p53 codon alignment, the
tree and the genetic code, build the MG94 model with the PAML estimates quoted there
(:22-27) and sum the per-column log-likelihoods -- one batched call instead of a
Python loop of 393 single-site calls.

    python examples/p53_loglik.py [alignment.phylip tree.newick genetic.code.txt]

Defaults to the copies of the reference's data under tests/golden/p53/.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from raoteh_amd import io, _mjp_dense      # noqa: E402


def main(argv):
    data = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'p53')
    aln = argv[1] if len(argv) > 1 else os.path.join(data, 'alignment.for.codeml.phylip')
    tree = argv[2] if len(argv) > 2 else os.path.join(data, 'p53S.const.tree')
    code_path = argv[3] if len(argv) > 3 else os.path.join(data, 'universal.code.txt')

    code = io.read_genetic_code(code_path)
    Q, distn = io.mg94_from_code(
        code, kappa=3.17632, omega=0.21925,
        nt_freqs=dict(A=0.25039, C=0.30126, G=0.25952, T=0.18883))
    T, root, leaf_name_pairs = io.read_newick(open(tree).read())
    leaves, states = io.alignment_to_states(io.read_phylip(aln), code, leaf_name_pairs)
    print('%d taxa, %d codon columns, %d distinct patterns, %d states' % (
        states.shape[1], states.shape[0], len(io.compress_patterns(states)[0]), len(code)))
    t0 = time.time()
    ll, status = _mjp_dense.get_log_likelihoods(
        T, root, len(code), leaves, states, kind='state', root_distn=distn,
        Q_default=Q, compress=True)
    print('total log likelihood: %.10f  (%d zero-probability columns, %.3f s)' % (
        ll.sum(), int((status & 1).sum()), time.time() - t0))


if __name__ == '__main__':
    main(sys.argv)
