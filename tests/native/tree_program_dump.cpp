// Test helper (host only): reads a tree from stdin (ntaxa nnodes, then parent / branch length / leaf taxon per node),
// compiles it with tapir_amd/csrc/tree_program.hpp and prints the plain and the fused op streams.
#include <cstdio>
#include <vector>
#include "tree_program.hpp"

int main() {
    int ntaxa = 0, nnodes = 0;
    if (scanf("%d %d", &ntaxa, &nnodes) != 2) return 2;
    std::vector<int32_t> parent(nnodes), leaf(nnodes);
    std::vector<double> blen(nnodes);
    for (int i = 0; i < nnodes; ++i) if (scanf("%d %lf %d", &parent[i], &blen[i], &leaf[i]) != 3) return 2;
    tphip::TreeProgram prog;
    const std::string err = tphip::build_tree_program(ntaxa, nnodes, parent.data(), blen.data(), leaf.data(), &prog);
    if (!err.empty()) { printf("error %s\n", err.c_str()); return 1; }
    printf("depth %d\n", prog.stack_depth);
    for (const auto& op : prog.ops) printf("plain %d %d %.17g\n", op.code, op.taxon, op.t);
    for (const auto& op : prog.fused_ops) printf("fused %d %d %.17g\n", op.code, op.taxon, op.t);
    return 0;
}
