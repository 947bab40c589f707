// Shared between pda_bpr_plan.hip (the step) and pda_bpr_plan_large.hip (the plan of a large batch).
#pragma once
// Very long segments of a LARGE batch (the hot positives of a popularity-skewed catalogue: one item carried 3 132 of the 65 536
// references of a Zipf batch of 32 768 triplets, and its one workgroup 70 of the step's 90 us): summed by kXlPieces workgroups,
// each over a fixed slice of the segment, the partial sums combined IN PIECE ORDER by the workgroup that arrives last (a counter
// per segment; no float atomics, the result does not depend on who arrives when).  pda_triplet_plan_large lists them.
constexpr int kXlMin = 512;          // references
constexpr int kXlPieces = 16;
