/* Material.hpp -- R/src/Material.h:10-34.  The constructor leaves diffuse / specular at zero and
 * does not touch shininess; containers value-initialise it to 0, which is what scenes that never
 * set a mesh material render with. */
#pragma once
#include "Vec3.hpp"

enum MaterialType { Material_Diffuse_Blinn_Phong, Material_Glass, Material_Mirror };

struct Material {
    Vec3 ambient_material;
    Vec3 diffuse_material;
    Vec3 specular_material;
    double shininess = 0.0;
    float index_medium = 1.0f;
    float transparency = 0.0f;
    MaterialType type = Material_Diffuse_Blinn_Phong;
};
