// mca::ArrayDescription -- microphone geometry carrier at the boundary.  Same public interface as the
// reference (include/mcarray/ArrayDescription.h:31-92, src/mcarray/ArrayDescription.cpp); plain host C++.
// Kept quirk: minDistance() starts its search at 0 and therefore always returns 0 (ArrayDescription.cpp:93-107).
#ifndef MCA_HIP_ARRAYDESCRIPTION_H
#define MCA_HIP_ARRAYDESCRIPTION_H
#include <algorithm>
#include <cmath>
#include <map>
#include <ostream>
#include <string>
#include <tuple>
#include <vector>

#include "mcarray_exception.h"

namespace mca {

class ArrayDescription {
public:
    typedef std::tuple<double, double, double> ArrayPosition;   // x, y, z in metres
    typedef int ElementId;

    ArrayDescription() {}
    virtual ~ArrayDescription() {}

    ElementId pushPosition(double x, double y, double z, const std::string &name = "")
    {
        return pushPosition(ArrayPosition(x, y, z), name);
    }
    ElementId pushPosition(const ArrayPosition &position, const std::string &name = "")
    {
        const ElementId id = static_cast<ElementId>(pos_.size());     // ids are dense 0..size-1
        const std::string key = name.empty() ? std::to_string(id) : name;
        if (names_.count(key)) throw MCArrayException("Array element name aready used");   // ArrayDescription.cpp:119
        names_[key] = id;
        pos_.push_back(position);
        return id;
    }
    size_t size() const { return pos_.size(); }
    bool empty() const { return pos_.empty(); }

    double distance(ElementId i, ElementId j) const
    {
        const ArrayPosition &a = at(i), &b = at(j);
        return std::sqrt(std::pow(std::get<0>(b) - std::get<0>(a), 2) + std::pow(std::get<1>(b) - std::get<1>(a), 2) +
                         std::pow(std::get<2>(b) - std::get<2>(a), 2));
    }
    double distance(const std::string &i, const std::string &j) const { return distance(getId(i), getId(j)); }
    double maxDistance() const
    {
        double m = 0;
        for (size_t i = 0; i < pos_.size(); ++i)
            for (size_t j = 0; j < pos_.size(); ++j)
                if (i != j) m = std::max(m, distance(static_cast<ElementId>(i), static_cast<ElementId>(j)));
        return m;
    }
    double minDistance() const
    {
        double m = 0;
        for (size_t i = 0; i < pos_.size(); ++i)
            for (size_t j = 0; j < pos_.size(); ++j)
                if (i != j) m = std::min(m, distance(static_cast<ElementId>(i), static_cast<ElementId>(j)));
        return m;
    }
    void getPosition(ElementId id, ArrayPosition &position) const { position = at(id); }
    void getPosition(const std::string &name, ArrayPosition &position) const { position = at(getId(name)); }
    double getX(const ElementId &id) const { return std::get<0>(at(id)); }
    double getY(const ElementId &id) const { return std::get<1>(at(id)); }
    double getZ(const ElementId &id) const { return std::get<2>(at(id)); }
    void getX(std::vector<double> &x) const { x.clear(); for (const auto &p : pos_) x.push_back(std::get<0>(p)); }
    void getY(std::vector<double> &y) const { y.clear(); for (const auto &p : pos_) y.push_back(std::get<1>(p)); }
    void getZ(std::vector<double> &z) const { z.clear(); for (const auto &p : pos_) z.push_back(std::get<2>(p)); }
    double getX(const std::string &name) const { return getX(getId(name)); }
    double getY(const std::string &name) const { return getY(getId(name)); }
    double getZ(const std::string &name) const { return getZ(getId(name)); }
    std::string getName(ElementId id) const
    {
        for (const auto &kv : names_) if (kv.second == id) return kv.first;
        return "";
    }
    ElementId getId(const std::string &name) const
    {
        auto it = names_.find(name);
        return it == names_.end() ? -1 : it->second;
    }
    double getBandwidth() const   // speed of sound / (2 * aperture), ArrayDescription.cpp:294-301
    {
        const double d = maxDistance();
        return d <= 0 ? 0 : 346.1 / (2 * d);
    }
    static ArrayDescription make_linear_array_description(const std::vector<double> &x)
    {
        ArrayDescription desc;
        for (double v : x) desc.pushPosition(v, 0, 0);
        return desc;
    }
    // [M][3] coordinates in id order, the form the C ABI takes (mca_hip_config::mic_xyz)
    std::vector<double> xyz() const
    {
        std::vector<double> out;
        for (const auto &p : pos_) { out.push_back(std::get<0>(p)); out.push_back(std::get<1>(p)); out.push_back(std::get<2>(p)); }
        return out;
    }

private:
    const ArrayPosition &at(ElementId id) const
    {
        if (id < 0 || static_cast<size_t>(id) >= pos_.size()) throw MCArrayException("No element with that id in the array description");
        return pos_[static_cast<size_t>(id)];
    }
    std::map<std::string, ElementId> names_;
    std::vector<ArrayPosition> pos_;
};

inline std::ostream &operator<<(std::ostream &os, const ArrayDescription &d)
{
    for (size_t i = 0; i < d.size(); ++i) {
        const int id = static_cast<int>(i);
        os << "{" << d.getName(id) << ": [" << d.getX(id) << ", " << d.getY(id) << ", " << d.getZ(id) << "]}  ";
    }
    return os;
}

}  // namespace mca
#endif
