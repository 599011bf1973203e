"""
Input formats of the reference's examples: sequential-paragraph PHYLIP codon
alignments, newick trees and the genetic-code table
(examples/p53/app_helper.py:80-139,158-183), the MG94 codon model builder
(examples/p53/create_mg94.py:23-142), the rate-matrix text format of
craoteh/README.rst:1-10, and site-pattern compression.  Host-side
only; the parsers are written from the formats, not from the reference's
dendropy-based readers.
"""
from __future__ import annotations

import itertools

import networkx as nx
import numpy as np

__all__ = ['read_phylip', 'read_newick', 'read_genetic_code', 'mg94_from_code',
           'alignment_to_states', 'compress_patterns', 'read_rate_matrix', 'write_rate_matrix']


def read_phylip(path_or_file):
    """-> list of (taxon name, list of codons).  A header line `ntaxa nsites`,
    then one blank-line-separated paragraph per taxon; nucleotides are grouped
    into codons in reading order."""
    fin = open(path_or_file) if isinstance(path_or_file, str) else path_or_file
    try:
        lines = [line.strip() for line in fin]
    finally:
        if isinstance(path_or_file, str):
            fin.close()
    paragraphs, para = [], []
    for line in lines:
        if line:
            para.append(line)
        elif para:
            paragraphs.append(para)
            para = []
    if para:
        paragraphs.append(para)
    header = paragraphs[0][0].split()
    ntaxa, nsites = int(header[0]), int(header[1])
    body = paragraphs[1:] if len(paragraphs[0]) == 1 else [paragraphs[0][1:]] + paragraphs[1:]
    if len(body) != ntaxa:
        raise ValueError('expected %d taxa, found %d paragraphs' % (ntaxa, len(body)))
    out = []
    for para in body:
        # name on its own line followed by sequence lines (the reference's `testseq`
        # layout), or `name  SEQUENCE...` starting on the name line (the shipped
        # alignment.for.codeml.phylip)
        tokens = para[0].split()
        name = tokens[0]
        seq = ''.join(tokens[1:] + ''.join(para[1:]).split()).upper()
        if len(seq) != nsites:
            raise ValueError('taxon %s: %d sites, expected %d' % (name, len(seq), nsites))
        if nsites % 3:
            raise ValueError('a codon alignment needs a multiple of 3 sites')
        out.append((name, [seq[i:i + 3] for i in range(0, nsites, 3)]))
    return out


def read_newick(text):
    """-> (T, root, leaf_name_pairs): undirected nx.Graph with edge 'weight',
    the root node and (node, name) for the leaves.  Node numbering follows
    examples/p53/app_helper.py:read_newick: leaves first in reading order, then
    the internal nodes in post-order, the root last.  Supports names, branch
    lengths and nested parentheses (no quoted labels, no comments)."""
    s = ''.join(text.split())
    if not s.endswith(';'):
        raise ValueError('a newick string ends with a semicolon')
    pos = [0]
    leaves, internal = [], []          # node records in creation order

    def parse():
        node = {'children': [], 'name': None, 'length': None}
        if s[pos[0]] == '(':
            pos[0] += 1
            while True:
                node['children'].append(parse())
                if s[pos[0]] == ',':
                    pos[0] += 1
                    continue
                if s[pos[0]] == ')':
                    pos[0] += 1
                    break
                raise ValueError('bad newick at position %d' % pos[0])
        start = pos[0]
        while s[pos[0]] not in ',():;':
            pos[0] += 1
        node['name'] = s[start:pos[0]] or None
        if s[pos[0]] == ':':
            pos[0] += 1
            start = pos[0]
            while s[pos[0]] not in ',();':
                pos[0] += 1
            node['length'] = float(s[start:pos[0]])
        (internal if node['children'] else leaves).append(node)   # post-order for internals
        return node

    top = parse()
    if s[pos[0]] != ';':
        raise ValueError('trailing characters in the newick string')
    index = dict((id(n), i) for i, n in enumerate(leaves + internal))
    T = nx.Graph()
    T.add_nodes_from(range(len(index)))
    for n in internal:
        for c in n['children']:
            T.add_edge(index[id(n)], index[id(c)],
                       weight=1.0 if c['length'] is None else c['length'])
    return T, index[id(top)], [(i, n['name']) for i, n in enumerate(leaves)]


def read_genetic_code(path_or_file):
    """-> list of (state, residue, codon) for the sense codons, in file order
    (examples/p53/app_helper.py:158-183: `state residue codon` per line, stop
    codons dropped)."""
    fin = open(path_or_file) if isinstance(path_or_file, str) else path_or_file
    try:
        code = []
        for line in fin:
            if line.strip():
                state, residue, codon = line.split()
                if residue.upper() != 'STOP':
                    code.append((int(state), residue.upper(), codon.upper()))
        return code
    finally:
        if isinstance(path_or_file, str):
            fin.close()


def mg94_from_code(genetic_code, kappa, omega, nt_freqs, target_expected_rate=1.0):
    """Muse-Gaut 94 rate matrix over the codons of `genetic_code` (triples as
    read_genetic_code returns them, states 0..n-1 in order): codons differing in
    exactly one nucleotide move at pi[target nt] * (kappa if transition) *
    (omega if the amino acid changes) (examples/p53/create_mg94.py:60-104);
    stationary distribution = normalised product of nucleotide frequencies;
    scaled so that the expected rate is `target_expected_rate` (:106-118).
    nt_freqs = dict or (A, C, G, T).  -> (Q f64[n,n], distn f64[n])."""
    if not isinstance(nt_freqs, dict):
        nt_freqs = dict(zip('ACGT', nt_freqs))
    states = [s for s, _, _ in genetic_code]
    if states != list(range(len(states))):
        raise ValueError('states must be 0..n-1 in order')
    n = len(states)
    transitions = {('A', 'G'), ('G', 'A'), ('C', 'T'), ('T', 'C')}
    Q = np.zeros((n, n))
    for (sa, ra, ca), (sb, rb, cb) in itertools.permutations(genetic_code, 2):
        diff = [(x, y) for x, y in zip(ca, cb) if x != y]
        if len(diff) != 1:
            continue
        x, y = diff[0]
        rate = nt_freqs[y]
        if (x, y) in transitions:
            rate *= kappa
        if ra != rb:
            rate *= omega
        Q[sa, sb] = rate
    w = np.array([nt_freqs[c[0]] * nt_freqs[c[1]] * nt_freqs[c[2]]
                  for _, _, c in genetic_code])
    distn = w / w.sum()
    Q -= np.diag(Q.sum(axis=1))
    expected_rate = -float(np.dot(distn, np.diag(Q)))
    Q *= target_expected_rate / expected_rate
    return Q, distn


def alignment_to_states(name_codons, genetic_code, leaf_name_pairs):
    """-> (leaves, states uint8[nsites, nleaves]): column order = leaves, 255 for
    a codon that is not a sense codon of the table (gaps, ambiguity: unobserved)."""
    codon_to_state = dict((c, s) for s, _, c in genetic_code)
    name_to_leaf = dict((name, leaf) for leaf, name in leaf_name_pairs)
    leaves = [name_to_leaf[name] for name, _ in name_codons]
    nsites = len(name_codons[0][1])
    states = np.full((nsites, len(leaves)), 255, dtype=np.uint8)
    for k, (_, codons) in enumerate(name_codons):
        for i, codon in enumerate(codons):
            states[i, k] = codon_to_state.get(codon.upper(), 255)
    return leaves, states


def compress_patterns(data):
    """Unique site patterns: -> (unique rows, inverse index int64[nsites], counts
    int64[npatterns]) with data[i] == unique[inverse[i]].  Works for the state
    (uint8), mask (uint64) and dense (f64, compared bitwise) encodings."""
    data = np.ascontiguousarray(data)
    flat = data.reshape(data.shape[0], -1)
    view = flat.view(np.dtype((np.void, flat.dtype.itemsize * flat.shape[1]))).ravel()
    _, first, inverse, counts = np.unique(view, return_index=True, return_inverse=True,
                                          return_counts=True)
    return data[first], inverse.astype(np.int64), counts.astype(np.int64)


def read_rate_matrix(path_or_file):
    """The primary-process rate-matrix text format of craoteh/README.rst:1-10: the first
    line holds the number of states N (states 0 .. N-1), every further non-empty line a
    whitespace-separated triple `source_state sink_state rate`; missing entries are zero
    and the diagonal is determined by the rows summing to zero.  -> f64[N, N].
    ValueError: a state out of range, a negative or non-finite rate, a diagonal or
    repeated entry, a malformed line."""
    fin = open(path_or_file) if isinstance(path_or_file, str) else path_or_file
    try:
        lines = [ln.split('#', 1)[0].strip() for ln in fin]
    finally:
        if isinstance(path_or_file, str):
            fin.close()
    lines = [ln for ln in lines if ln]
    if not lines:
        raise ValueError('empty rate matrix file')
    head = lines[0].split()
    if len(head) != 1 or not head[0].isdigit() or int(head[0]) < 1:
        raise ValueError('the first line must hold the number of states')
    n = int(head[0])
    Q = np.zeros((n, n), dtype=np.float64)
    seen = set()
    for ln in lines[1:]:
        parts = ln.split()
        if len(parts) != 3:
            raise ValueError('expected `source sink rate`, got %r' % ln)
        try:
            a, b, r = int(parts[0]), int(parts[1]), float(parts[2])
        except ValueError:
            raise ValueError('expected `source sink rate`, got %r' % ln)
        if not (0 <= a < n and 0 <= b < n):
            raise ValueError('state out of range in %r' % ln)
        if a == b:
            raise ValueError('diagonal entries are determined automatically: %r' % ln)
        if not np.isfinite(r) or r < 0:
            raise ValueError('rates must be finite and non-negative: %r' % ln)
        if (a, b) in seen:
            raise ValueError('repeated entry %r' % ln)
        seen.add((a, b))
        Q[a, b] = r
    Q[np.arange(n), np.arange(n)] = -Q.sum(axis=1)
    return Q


def write_rate_matrix(Q, path_or_file):
    """Inverse of read_rate_matrix: N, then one tab-separated triple per nonzero
    off-diagonal rate (repr of the float: the file reads back bit for bit)."""
    Q = np.asarray(Q, dtype=np.float64)
    if Q.ndim != 2 or Q.shape[0] != Q.shape[1]:
        raise ValueError('expected the array to be square')
    fout = open(path_or_file, 'w') if isinstance(path_or_file, str) else path_or_file
    try:
        fout.write('%d\n' % Q.shape[0])
        for a, b in zip(*np.nonzero(Q)):
            if a != b:
                fout.write('%d\t%d\t%r\n' % (a, b, float(Q[a, b])))
    finally:
        if isinstance(path_or_file, str):
            fout.close()
