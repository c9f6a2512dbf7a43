// ORACLE SUPPORT: compiles the reference's src/ik_evolution_2.cpp, unmodified, from where it lies
#include "ref_prelude.h"
#include "ik_evolution_2.cpp"

// Read-only window onto the reference solver's population, for step-level comparisons (the class is local to the
// reference's translation unit, so the accessor has to live in this one).
namespace {
template <int M>
bool dump(bio_ik::IKBase* b, double* genes, double* fitness, double* solution_fitness) {
    auto* e = dynamic_cast<bio_ik::IKEvolution2<M>*>(b);
    if (!e) return false;
    size_t D = e->problem.active_variables.size();
    for (size_t s = 0; s < 2; s++) {
        for (size_t i = 0; i < 2; i++)
            for (size_t g = 0; g < D; g++) {
                genes[((s * 2 + i) * 2 + 0) * D + g] = e->species[s].individuals[i].genes[g];
                genes[((s * 2 + i) * 2 + 1) * D + g] = e->species[s].individuals[i].gradients[g];
            }
        fitness[s] = e->species[s].fitness;
    }
    *solution_fitness = e->solution_fitness;
    return true;
}
}  // namespace
extern "C" int ref_evolution_state(void* ikbase, double* genes, double* fitness, double* solution_fitness) {
    auto* b = (bio_ik::IKBase*)ikbase;
    return (dump<0>(b, genes, fitness, solution_fitness) || dump<'q'>(b, genes, fitness, solution_fitness) || dump<'l'>(b, genes, fitness, solution_fitness)) ? 0 : -1;
}
