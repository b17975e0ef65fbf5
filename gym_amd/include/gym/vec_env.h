// vec_env.h — stand-alone mirror of the reference's batch interface, for trees that do not carry the reference headers.
//
// Same names, members and meaning as gym::VecEnv<dict> / VecEnvStepReturn / VecEnvWrapper<dict>
// (wrappers/vec_env/base.h:18-79, vec_env_wrapper.h:12-55) and the space types the interface exposes
// (spaces/space.h:20-104, box.h:15-54, discrete.h:14-31, util.h:18-26, common/utils.h:82-122).  Inside the reference tree
// define GYMHIP_USE_REFERENCE_HEADERS before including hip_vec_env.h and these definitions are skipped in favour of the
// reference's own.
#pragma once
#ifndef GYMHIP_USE_REFERENCE_HEADERS

#include <any>
#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include <torch/torch.h>

namespace gym {

using AnyMap = std::unordered_map<std::string, std::any>;   // env.h:22
using TensorDict = std::unordered_map<std::string, torch::Tensor>;

struct Result { float r, l, t; };   // monitor.h:16-18: info["episode"]

namespace space {

struct Shape { std::vector<int64_t> m_Axis{}; c10::ScalarType m_Type{c10::kInt}; };   // util.h:18-21
template <typename T> struct Range { T low{}, high{}; };                               // util.h:23-26

class Space {   // space.h:20-104 (sampling lives with the caller: the reference seeds it from random_device)
public:
    Space() = default;
    explicit Space(Shape s) : m_Shape(std::move(s)) {}
    virtual ~Space() = default;
    template <typename T> T* as() { return dynamic_cast<T*>(this); }
    virtual std::shared_ptr<Space> clone() = 0;
    auto type() const { return m_Shape.m_Type; }
    auto size() const { return m_Shape.m_Axis; }
protected:
    Shape m_Shape{};
};

template <typename T> class Box : public Space {   // box.h:15-54
public:
    Box(std::vector<Range<T>> range, Shape const& shape) : Space(shape), m_Range(std::move(range)) {
        if (m_Range.empty()) throw std::invalid_argument("Box: empty range");   // box.tpp:10-17
    }
    explicit Box(Shape const& shape) : Space(shape), m_Range({Range<T>{std::numeric_limits<T>::min(), std::numeric_limits<T>::max()}}) {}
    std::vector<Range<T>> getRange() const { return m_Range; }
    std::shared_ptr<Space> clone() override { return std::make_shared<Box<T>>(*this); }
protected:
    std::vector<Range<T>> m_Range{};
};

class Discrete : public Space {   // discrete.h:14-31
public:
    const int64_t n;
    explicit Discrete(int64_t _n) : Space(Shape{{1}, c10::kInt}), n(_n) {
        if (_n <= 0) throw std::runtime_error("Discrete: n must be positive");   // discrete.cpp:19-20
    }
    std::shared_ptr<Space> clone() override { return std::make_shared<Discrete>(*this); }
};

}  // namespace space

using namespace space;

template <typename ObservationType> struct VecEnvStepReturn {   // base.h:19-25
    ObservationType observation{};
    torch::Tensor reward{};
    torch::Tensor done{};
    std::vector<AnyMap> info;
};

template <bool dict> using infer_type = std::conditional_t<dict, TensorDict, torch::Tensor>;

template <bool dict> class VecEnv {   // base.h:30-79
public:
    using ObservationT = infer_type<dict>;
    using ActionT = torch::Tensor;
    using StepT = VecEnvStepReturn<ObservationT>;

    VecEnv() = default;
    virtual ~VecEnv() = default;
    VecEnv(int numEnvs, std::shared_ptr<Space> const& o_space, std::shared_ptr<Space> const& a_space)
        : numEnvs(numEnvs), m_ObservationSpace(o_space), m_ActionSpace(a_space) {}

    auto actionSpace() const noexcept { return m_ActionSpace; }
    auto observationSpace() const noexcept { return m_ObservationSpace; }
    auto nEnvs() const { return numEnvs; }

    virtual ObservationT reset() noexcept = 0;
    virtual void stepAsync(torch::Tensor const& actions) noexcept = 0;
    virtual StepT stepWait() noexcept = 0;
    StepT step(torch::Tensor const& actions) noexcept { stepAsync(actions); return stepWait(); }
    virtual bool isWrapper() const { return false; }
    virtual ObservationT reset(int index) = 0;
    virtual StepT step(int index, torch::Tensor const& action) noexcept = 0;
    virtual void render() const = 0;
    virtual std::vector<size_t> seed(size_t) noexcept { return {}; }

protected:
    int numEnvs{};
    std::shared_ptr<Space> m_ObservationSpace, m_ActionSpace;
};

}  // namespace gym

#endif  // GYMHIP_USE_REFERENCE_HEADERS
