// stand-in: types named by TouchGoal / BalanceGoal declarations (never instantiated on the hot path)
#pragma once
#include <memory>
namespace shapes {
struct Shape {};
struct Mesh : Shape {};
typedef std::shared_ptr<const Shape> ShapeConstPtr;
}  // namespace shapes
