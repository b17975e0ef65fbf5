// vec_env.h — stand-alone mirror of the reference's batch interface, for trees that do not carry the reference headers.
//
// Same names, members and meaning as gym::VecEnv<dict> / VecEnvStepReturn / VecEnvWrapper<dict>
// (wrappers/vec_env/base.h:18-79, vec_env_wrapper.h:12-55), the single-env gym::Env face (env.h:26-85) and the space types the
// interface exposes (spaces/space.h:20-104, box.h:15-54, box.tpp:7-18, discrete.h:14-31, discrete.cpp:15-21, util.h:18-97,
// common/utils.h:82-122).  Inside the reference tree define GYMHIP_USE_REFERENCE_HEADERS before including hip_vec_env.h and
// these definitions are skipped in favour of the reference's own (INTEGRATION.md shows that configuration; the build container compiles it
// against the reference's headers where they lie).  tests/test_ref_pin.py checks this mirror's space types
// against values the reference's own spaces code produced (tests/golden/ref_spaces.json).
#pragma once
#ifndef GYMHIP_USE_REFERENCE_HEADERS

#include <any>
#include <cstdint>
#include <iostream>
#include <limits>
#include <memory>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include <torch/torch.h>

namespace gym {

using AnyMap = std::unordered_map<std::string, std::any>;   // env.h:22
using ArgMap = std::unordered_map<std::string, std::any>;   // env.h:23
using Kwargs = std::optional<ArgMap>;                        // env.h:24
using TensorDict = std::unordered_map<std::string, torch::Tensor>;

struct Result { float r, l, t; };   // monitor.h:16-18: info["episode"]

namespace space {

struct Shape { std::vector<int64_t> m_Axis{}; c10::ScalarType m_Type{c10::kInt}; };   // util.h:18-21
template <typename T> struct Range { T low{}, high{}; };                               // util.h:23-26
inline size_t dim(Shape const& shape) { return shape.m_Axis.size(); }                 // util.h:34-37

// util.h:28-32,58-102.  uint8_t maps to c10::kChar (int8) in the reference (util.h:60-63,80-83) — a quirk kept on purpose:
// code that asks a space for its type must see what the reference's spaces report.
template <typename T> inline Shape make_shape(std::vector<int64_t> const& axis);
template <typename T> inline Shape make_shape(int64_t vector_length) { return make_shape<T>(std::vector<int64_t>{vector_length}); }
template <> inline Shape make_shape<int>(std::vector<int64_t> const& axis) { return {axis, c10::kInt}; }
template <> inline Shape make_shape<uint8_t>(std::vector<int64_t> const& axis) { return {axis, c10::kChar}; }
template <> inline Shape make_shape<float>(std::vector<int64_t> const& axis) { return {axis, c10::kFloat}; }
template <> inline Shape make_shape<double>(std::vector<int64_t> const& axis) { return {axis, c10::kDouble}; }
template <> inline Shape make_shape<bool>(int64_t vector_length) { return {{vector_length}, c10::kBool}; }   // only the length form exists (util.h:75-78)

class Space {   // space.h:20-104
public:
    Space() = default;
    explicit Space(Shape s) : m_Shape(std::move(s)) {   // space.h:64-67: the sampler is seeded from std::random_device
        std::random_device rd;
        m_Generator = std::mt19937{rd()};
    }
    virtual ~Space() = default;
    template <typename T> T sample();                                                   // space.h:69-70, space.cpp:11-33
    template <typename T> T* as() { return dynamic_cast<T*>(this); }                    // space.h:72-74
    virtual std::vector<std::string> keys() const { return {m_Name}; }                  // space.h:75-76
    void setName(std::string const& name) { m_Name = name; }
    virtual std::shared_ptr<Space> clone() = 0;
    auto type() const { return m_Shape.m_Type; }
    auto size() const { return m_Shape.m_Axis; }
protected:
    const Shape m_Shape{};
    std::mt19937 m_Generator{};
    std::string m_Name{"observation"};
};

template <typename T> class Box : public Space {   // box.h:15-54
public:
    Box(Range<T> range, Shape const& shape) : Space(shape), m_Range({range}) {}          // box.h:22-25
    Box(std::vector<Range<T>> range, Shape const& shape) : Space(shape), m_Range(std::move(range)) {   // box.tpp:7-18
        if (dim(shape) == 0) throw std::runtime_error("a zero dim m_Shape or scalar m_Shape is not allowed");
        if (dim(shape) > 1) throw std::runtime_error("Initializing a Box m_Shape with a vectorized range requires that a m_Shape has only one dim");
    }
    // box.h:31-34: ONE range {numeric_limits<T>::min(), max()} — for float that is the smallest POSITIVE value (a reference quirk)
    explicit Box(Shape const& shape) : Space(shape), m_Range({Range<T>{std::numeric_limits<T>::min(), std::numeric_limits<T>::max()}}) {}
    Box(Box&&) = default;
    Box(Box const&) = default;
    std::vector<Range<T>> getRange() const { return m_Range; }
    void getRange(std::vector<T>& low, std::vector<T>& high) const {                     // box.h:40-46
        low.clear(); high.clear();
        for (auto const& r : m_Range) { low.push_back(r.low); high.push_back(r.high); }
    }
    std::shared_ptr<Space> clone() override { return std::make_shared<Box<T>>(*this); }
protected:
    std::vector<Range<T>> m_Range{};
};

class Discrete : public Space {   // discrete.h:14-31
public:
    const int64_t n;
    explicit Discrete(int64_t _n) : Space(Shape{{_n}, c10::kInt}), n(_n) {              // discrete.cpp:15-21: the shape is {n}
        if (_n <= 0) throw std::invalid_argument("discrete space only accepts dim > 0");
    }
    std::shared_ptr<Space> clone() override { return std::make_shared<Discrete>(*this); }
};

template <> inline std::vector<float> Space::sample() {   // space.cpp:11-27
    std::vector<float> res;
    if (auto box = dynamic_cast<Box<float>*>(this)) {
        for (auto& range : box->getRange()) {
            std::uniform_real_distribution<float> distribution(range.low, range.high);
            res.emplace_back(distribution(m_Generator));
        }
    } else {
        std::cerr << "Error samplying float: only allowed for box environment\n";
    }
    return res;
}
template <> inline int Space::sample() {   // space.cpp:29-34
    std::uniform_int_distribution<int64_t> distribution(0, int(m_Shape.m_Axis[0]) - 1);
    return static_cast<int>(distribution(m_Generator));
}

}  // namespace space

using namespace space;

// common/utils.h:82-122 (that header also drags OpenCV in; only the space factories are mirrored)
template <typename T> static inline std::unique_ptr<space::Space> makeBoxSpace(space::Shape const& shape) { return std::make_unique<space::Box<T>>(shape); }
template <typename T> static inline std::unique_ptr<space::Space> makeBoxSpace(int64_t vector_length) { return std::make_unique<space::Box<T>>(make_shape<T>(vector_length)); }
template <typename T> static inline std::unique_ptr<space::Space> makeBoxSpace(std::vector<int64_t> shape) { return std::make_unique<space::Box<T>>(make_shape<T>(shape)); }
template <typename T> static inline std::unique_ptr<space::Space> makeBoxSpace(std::vector<T> lows, std::vector<T> highs, int64_t vector_length) {
    if (highs.size() < lows.size()) throw std::invalid_argument("makeBoxSpace: fewer highs than lows");   // the reference reads past the end here (utils.h:103-107)
    std::vector<Range<T>> rangeVector(lows.size());
    for (size_t i = 0; i < lows.size(); i++) rangeVector[i] = Range<T>{lows[i], highs[i]};
    return std::make_unique<space::Box<T>>(rangeVector, make_shape<T>(vector_length));
}
template <typename T> static inline std::unique_ptr<space::Space> makeBoxSpace(T lows, T highs, std::vector<int64_t> const& sizes) {
    return std::make_unique<space::Box<T>>(Range<T>{lows, highs}, make_shape<T>(sizes));
}
static inline std::unique_ptr<space::Space> makeDiscreteSpace(size_t n) { return std::make_unique<space::Discrete>(n); }

// env.h:26-85 — the single-env face
template <typename ObservationType> struct StepResponse {   // env.h:26-32
    ObservationType observation{};
    float reward{0.0};
    bool done{false};
    AnyMap info{};
};

template <typename ObservationType, typename ActionType, typename StepType = StepResponse<ObservationType>> class Env {   // env.h:34-85
public:
    using StepT = StepType;
    using ActionT = ActionType;
    using ObservationT = ObservationType;
    using BaseEnvT = Env<ObservationType, ActionType, StepType>;
    explicit Env(Kwargs kwargs = std::nullopt) : m_Args(std::move(kwargs)) {}
    Env(Env const&) = default;
    Env(Env&&) noexcept = default;
    virtual ~Env() = default;
    virtual ObservationT reset() noexcept = 0;
    virtual void render() {}
    virtual std::string info() { return "Gym Env"; }
    virtual void seed(std::optional<uint64_t> const& _seed) noexcept {   // env.h:60-68
        if (_seed) m_Device.seed(*_seed);
        else { std::random_device r; std::seed_seq seed{r(), r()}; m_Device = std::mt19937(seed); }
    }
    virtual StepT step(ActionT const& action) noexcept = 0;
    [[nodiscard]] auto actionSpace() const noexcept { return m_ActionSpace; }
    [[nodiscard]] auto observationSpace() const noexcept { return m_ObservationSpace; }
    template <typename T> T getArg(std::string const& key, T default_) noexcept {
        return (!m_Args || m_Args->find(key) == m_Args->end()) ? default_ : std::any_cast<T>(m_Args->at(key));
    }
protected:
    std::shared_ptr<Space> m_ActionSpace, m_ObservationSpace;
    std::mt19937 m_Device;
    Kwargs m_Args{};
};

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

template <bool dict> class VecEnvWrapper : public VecEnv<dict> {   // vec_env_wrapper.h:12-55
public:
    VecEnvWrapper() = default;
    explicit VecEnvWrapper(std::unique_ptr<VecEnv<dict>> venv, std::shared_ptr<Space> o_space = nullptr, std::shared_ptr<Space> a_space = nullptr)
        : VecEnv<dict>(venv->nEnvs(), std::move(o_space), std::move(a_space)) {
        this->venv = std::move(venv);
        if (!this->m_ObservationSpace) this->m_ObservationSpace = this->venv->observationSpace();
        if (!this->m_ActionSpace) this->m_ActionSpace = this->venv->actionSpace();
    }
    typename VecEnv<dict>::ObservationT reset() noexcept override { return venv->reset(); }
    void stepAsync(torch::Tensor const& actions) noexcept override { this->venv->stepAsync(actions); }
    void render() const override { this->venv->render(); }
    typename VecEnv<dict>::StepT stepWait() noexcept override { return venv->stepWait(); }
    std::vector<size_t> seed(size_t _seed) noexcept override { return venv->seed(_seed); }
    typename VecEnv<dict>::ObservationT reset(int index) override { return venv->reset(index); }
    typename VecEnv<dict>::StepT step(int index, torch::Tensor const& action) noexcept override { return venv->step(index, action); }
protected:
    std::unique_ptr<VecEnv<dict>> venv{nullptr};
};

// updateable.h:10-31: the observer a learner uses to push rollout tensors into VecNormalize
template <class... Args> struct Updateable {
    virtual void update(Args&&...) = 0;
};
template <class... Args> struct Updater {
    void subscribe(Updateable<Args...>* sub) {
        if (!sub) throw std::invalid_argument("Updater::subscribe(nullptr)");   // the reference asserts
        updateables.emplace_back(sub);
    }
    void notifyOne(Args&&... args) {   // only the FIRST subscriber is told (updateable.h:26-29)
        if (!updateables.empty()) updateables[0]->update(std::forward<Args>(args)...);
    }
private:
    std::vector<Updateable<Args...>*> updateables{};
};

template <bool dict> struct VecNormalizeStateRewardPair {   // vec_normalize.h:14-18
    infer_type<dict> state;
    torch::Tensor reward, discounts;
};

}  // namespace gym

#endif  // GYMHIP_USE_REFERENCE_HEADERS
