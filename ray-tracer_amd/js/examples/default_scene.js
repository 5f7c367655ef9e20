// Scene description in the reference's format: function generateScene(meshes) returning
// { objects, materials, lights, ambientLight, bgColor }. This one describes the 8-kd_tree demo layout (two
// spheres, checkered floor, cube, cylinder, one PLY mesh, two point lights) from a table; MESH_MATERIAL may be
// preset in the context to override the mesh's Phong constants (the benchmark gives it reflection and
// transmission).
function generateScene(meshes) {
    var V = function (x, y, z) { return new THREE.Vector3(x, y, z); };
    var lights = [
        new PointLight(nextPointLightId++, V(10, 300, -10), V(1, 1, 1)),
        new PointLight(nextPointLightId++, V(-300, 50, 200), V(0.1, 1, 0.1)),
    ];
    var mm = (typeof MESH_MATERIAL !== 'undefined' && MESH_MATERIAL) || [0.7, 0.6, 0.7, 16, 0, 0, 0];
    var M = {
        sphere1: new Phong(nextPhongId++, 0.075, 0.075, 0.2, 20, 0.01, 0.8, 0.95, V(1, 1, 1), V(1, 1, 1)),
        sphere2: new Phong(nextPhongId++, 0.15, 0.25, 1.0, 20, 0.75, 0, 0, V(0.7, 0.7, 0.7), V(1, 1, 1)),
        cube: new Phong(nextPhongId++, 0.5, 0.9, 0.2, 24, 0, 0, 0, V(0.8, 0.1, 0.7), V(0.7, 0.7, 0.7)),
        cyl: new Phong(nextPhongId++, 0.5, 0.3, 0.9, 20, 0, 0, 0, V(0.85, 0.7, 0.1), V(0.7, 0.7, 0.7)),
        floor: new PhongCheckered(nextPhongCheckeredId++, 0.5, 0.9, 0.2, 12, 0.4, 0, 0, V(1, 0, 0), V(1, 1, 0), V(0.7, 0.7, 0.7)),
        mesh: new Phong(nextPhongId++, mm[0], mm[1], mm[2], mm[3], mm[4], mm[5], mm[6], V(0.1, 0.75, 0.1), V(0.7, 0.7, 0.7)),
    };
    var table = [
        // class, scale, rotation (Euler XYZ) or null, translation, material
        [Sphere, V(50, 50, 50), null, V(2, 15, -240), M.sphere1],
        [Sphere, V(40, 40, 40), null, V(-75, -20, -325), M.sphere2],
        [Rectangle, V(450, 1, 1600), null, V(-105, -80, -300), M.floor],
        [Cube, V(90, 60, 60), new THREE.Euler(-Math.PI / 4, Math.PI / 4, 0, 'XYZ'), V(-70, 40, -400), M.cube],
        [Cylinder, V(30, 60, 30), new THREE.Euler(-Math.PI / 3, 0, -Math.PI / 6, 'XYZ'), V(50, 75, -300), M.cyl],
    ];
    var objects = table.map(function (row) {
        var o = new row[0](nextObjectId++);
        o.scale(row[1]);
        if (row[2]) o.rotate(row[2]);
        o.translate(row[3]);
        o.setMaterial(row[4]);
        return o;
    });
    g_verts = [];
    g_faces = [];
    if (meshes && meshes[MODEL_FILE_NAME]) {
        var mesh = new Mesh(nextObjectId++, meshes[MODEL_FILE_NAME].verts, meshes[MODEL_FILE_NAME].faces);
        mesh.scale(V(1000, 1000, 1000));
        mesh.translate(V(15, 27, -245));
        mesh.setMaterial(M.mesh);
        objects.push(mesh);
    }
    return {
        objects: objects,
        materials: [M.sphere1, M.sphere2, M.cube, M.cyl, M.floor, M.mesh],
        lights: lights,
        ambientLight: V(0.75, 0.75, 0.75),
        bgColor: V(0.1, 0.5, 0.9),
    };
}
